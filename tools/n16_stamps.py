#!/usr/bin/env python3
"""Diagnostic: in-kernel cycle stamps of the n16 edge launches of a config-2 step (pf_n16.hip built with -DN16_STAMPS as
variants/libpfdyn_n16stamps.so; see tools/n16_trace.py for the build recipe).  Stamps per wave: kernel entry | item known |
rows gathered | per block: start, [main issued, barrier A passed,] gate issued, barrier B passed | chain done.
    PFDYN_N16=2 PFDYN_LIB=.../variants/libpfdyn_n16stamps.so OFFS=0,64,128,300 python tools/n16_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pharmacoforge_amd as pfa  # noqa: E402
from pharmacoforge_amd import schedule, synthetic  # noqa: E402

dev = torch.device('cuda', 0)
B, T = int(os.environ.get("B", "32")), 500
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
lib = eng.lib
lib.pfk_n16_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int]
# KID: record one kernel of the step only (0 k_n16_edge<true>, 1 <false>, 2 k_n16_fused, 3 k_n16_tail; default: all, the last writer wins)
KID = int(os.environ.get("KID", "-1"))
if hasattr(lib, "pfk_n16_set_stamp_kernel"):
    lib.pfk_n16_set_stamp_kernel.argtypes = [ctypes.c_int]
    lib.pfk_n16_set_stamp_kernel(KID)
eng.prepare_timesteps(carr)            # (the shipped path: center tables and 'pa' rows computed ahead)
eng.sample_begin(noise[0])
for i in range(30):
    eng.denoise_step(carr[i], noise[i + 1])
torch.cuda.synchronize()
buf = torch.zeros(64 * 4 * 64, dtype=torch.int64, device=dev)
for off in [int(x) for x in os.environ.get("OFFS", "0").split(",")]:
    buf.zero_()
    assert lib.pfk_n16_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()), off) == 0
    eng.denoise_step(carr[30], noise[31])
    torch.cuda.synchronize()
    lib.pfk_n16_set_stamp_buffer(None, 0)
    st = buf.cpu().view(64, 4, 64).tolist()
    print(f"== workgroups from {off}: deltas between consecutive stamps of wave 0 (and the wave's total)")
    shown = 0
    for b in range(64):
        r = [x for x in st[b][0][:40] if x]
        if len(r) < 3:
            continue
        print(f"  wg {off + b:4d}: total {r[-1] - r[0]:6d} | " + " ".join(str(r[i + 1] - r[i]) for i in range(len(r) - 1)))
        sb = [(k, x) for k, x in enumerate(st[b][0][40:56]) if x]          # the tail launch's update + build phases (pf_stepbuild.h)
        if sb:
            order = sorted(sb, key=lambda kv: kv[1])
            print(f"           build: head done -> phase " + " ".join(f"{k}:+{x - r[-1]}" for k, x in order) + f" | kernel entry -> last {order[-1][1] - r[0]}")
        shown += 1
        if shown >= int(os.environ.get("SHOW", "5")):
            break
