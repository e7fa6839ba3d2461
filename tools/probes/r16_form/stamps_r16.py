#!/usr/bin/env python3
"""Diagnostic: in-kernel cycle stamps of k_r16_pp (pf_r16.hip) on a -DPF_STAMPS build:
    pharmacophore-diffusion_amd/csrc/build_variant.sh stamps "-DPF_STAMPS"
    PFDYN_LIB=$PWD/pharmacophore-diffusion_amd/csrc/variants/libpfdyn_stamps.so PFDYN_R16_ROWS_MIN=0 python tools/stamps_r16.py
Stamps per item: 0 work item known, 1 rows gathered, 2.. after each GVP block, 8 after the flush gates, 9 stores issued."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pharmacoforge_amd as pfa  # noqa: E402
from pharmacoforge_amd import schedule, synthetic  # noqa: E402

dev = torch.device('cuda', 0)
B, T = int(os.environ.get("B", "128")), 500
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
eng.sample_begin(noise[0])
for i in range(20):
    eng.denoise_step(carr[i], noise[i + 1])
torch.cuda.synchronize()
assert eng.l0_hoist() == 16, "run with PFDYN_R16_ROWS_MIN=0"
buf = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
lib = eng.lib
lib.pfk_r16_set_stamp_buffer.argtypes = [ctypes.c_void_p]
assert lib.pfk_r16_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
eng.denoise_step(carr[20], noise[21])
torch.cuda.synchronize()
lib.pfk_r16_set_stamp_buffer(None)
st = buf.cpu().view(64, 16)
print("k_r16_pp: cycles [gather | GVP blocks ... | flush gates | scans + stores]  total")
for w in range(0, 64, 8):
    r = st[w]
    ks = [k for k in range(10) if int(r[k]) != 0]
    print(f"  item {w:2d}: " + " ".join(str(int(r[ks[i + 1]] - r[ks[i]])) for i in range(len(ks) - 1)) + f"   total {int(r[ks[-1]] - r[ks[0]])}")
