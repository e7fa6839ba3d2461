#!/usr/bin/env python3
"""Diagnostic: in-kernel cycle stamps (s_memtime) of the KS row-group kernels (pf_rgk.hip), config 2.
    cd pharmacophore-diffusion_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPF_KS_STAMPS -c pf_rgk.hip -o /tmp/rgk_st.o && \
        hipcc -shared -fPIC --offload-arch=gfx950 pf_kernels.o pf_train.o pf_rg.o /tmp/rgk_st.o pf_r16.o pf_host.o -o variants/libpfdyn_ksst.so
    PFDYN_LIB=.../variants/libpfdyn_ksst.so PFDYN_KS_MAX=600 python tools/stamps_ks.py
The LAST KS launch of a denoising step (node update + noise head) leaves its stamps in the buffer."""
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
buf = torch.zeros(64 * 96, dtype=torch.int64, device=dev)
lib = eng.lib
lib.pfk_ks_set_stamp_buffer.argtypes = [ctypes.c_void_p]
eng.sample_begin(noise[0])
for i in range(30):
    eng.denoise_step(carr[i], noise[i + 1])
torch.cuda.synchronize()
assert lib.pfk_ks_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
eng.denoise_step(carr[30], noise[31])
torch.cuda.synchronize()
lib.pfk_ks_set_stamp_buffer(None)
st = buf.cpu().view(64, 96)
for slot in (0, 1, 2, 3, 4, 5):
    row = st[slot]
    n = int((row != 0).sum())
    dl = [int(row[i + 1] - row[i]) for i in range(0, n - 1)]
    print(f"wg {slot // 4} wave {slot % 4}: total {int(row[n - 1] - row[0])} cycles, {n} stamps: {dl}")
