#!/usr/bin/env python3
"""Diagnostic: in-kernel cycle stamps (s_memtime) of the cooperative GVP phases in the noise-head kernel.

Needs a diagnostic build of the kernels (never the product build):
    pharmacophore-diffusion_amd/csrc/build_variant.sh stamps "-DPF_STAMPS"
    PFDYN_LIB=$PWD/pharmacophore-diffusion_amd/csrc/variants/libpfdyn_stamps.so python tools/stamps_head.py
Prints, per wave of block 0, the cycles from kernel start to the first GVP and then, per GVP,
[vector products, main k-steps, wait at barrier 1, sh + SiLU + gates, wait at barrier 2, assemble -> next GVP]."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pharmacoforge_amd as pfa  # noqa: E402
from pharmacoforge_amd import schedule, synthetic  # noqa: E402

dev = torch.device('cuda', 0)
B, T = 32, 500
eng = pfa.PfEngine(device=dev)
eng.load_state_dict(synthetic.make_state_dict(0))
xs, hs = zip(*[synthetic.synthetic_pocket(1000 + i, 256) for i in range(B)])
px, ph = torch.cat(xs).to(dev), torch.cat(hs).to(dev)
pptr = torch.arange(B + 1) * 256
fptr = torch.arange(B + 1) * 6
s, d = eng.build_pp_edges(px, pptr)
eng.set_batch(px, ph, pptr, fptr, s, d)
coef = schedule.step_coefficients(schedule.PredefinedNoiseSchedule('polynomial_2', T, 1e-5).gamma, T)
carr = eng.coef_array(coef, list(range(39, -1, -1)))
noise = torch.randn(41, B * 6, 9, device=dev)
buf = torch.zeros(64 * 4 * 64, dtype=torch.int64, device=dev)
lib = eng.lib
lib.pfk_set_stamp_buffer.argtypes = [ctypes.c_void_p]
eng.sample_begin(noise[0])
for i in range(30):
    eng.denoise_step(carr[i], noise[i + 1])
torch.cuda.synchronize()
assert lib.pfk_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
eng.denoise_step(carr[30], noise[31])
torch.cuda.synchronize()
lib.pfk_set_stamp_buffer(None)
st = buf.cpu().view(64, 4, 64)          # the head kernel runs last, so its stamps are the ones left in slots 0..
for wv in range(4):
    row = st[0, wv]
    dl = [int(row[i + 1] - row[i]) for i in range(0, 26)]
    print(f"wave {wv}: start->gvp0 {dl[0]} | " + " | ".join(str(dl[1 + 6 * g: 7 + 6 * g]) for g in range(4)))
