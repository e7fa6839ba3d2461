#!/usr/bin/env python3
"""Diagnostic: in-kernel cycle stamps (s_memtime) of the row-group kernels (pf_rg.hip) at BASELINE config 2.

Needs a diagnostic build of the kernels (never the product build):
    pharmacophore-diffusion_amd/csrc/build_variant.sh stamps "-DPF_STAMPS"
    PFDYN_LIB=$PWD/pharmacophore-diffusion_amd/csrc/variants/libpfdyn_stamps.so python tools/stamps_rg.py
Every row-group launch of a step is stamped in turn (pfk_rg_set_stamp_which selects the launch); lane 0 of the first
64 waves writes s_memtime at the phase boundaries of rg_gvp / rg_flush (see the stamp(lane) comments in pf_rg.hip).
s_memtime ticks are shader cycles; the stamps themselves cost ~100-200 cycles each."""
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
buf = torch.zeros(64 * 64, dtype=torch.int64, device=dev)
lib = eng.lib
lib.pfk_rg_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.pfk_rg_set_stamp_which.argtypes = [ctypes.c_int]
eng.sample_begin(noise[0])
for i in range(30):
    eng.denoise_step(carr[i], noise[i + 1])
torch.cuda.synchronize()
names = {0: "k_rg_edge (conv layer 0)", 1: "k_rg_node (conv layer 0)", 2: "k_rg_edge (last layer)", 3: "k_rg_node + head (last layer)"}
if os.environ.get("PFDYN_N16", "3") != "0":          # the edge launches run on the n16 kernels (tools/n16_stamps.py): only the node launches are row-group launches
    names = {0: "k_rg_node (conv layer 0)", 1: "k_rg_node + head (last layer)", 2: "-", 3: "-"}
if hasattr(lib, "pfk_build_set_stamp_buffer"):
    bb = torch.zeros(B * 32, dtype=torch.int64, device=dev)
    lib.pfk_build_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    assert lib.pfk_build_set_stamp_buffer(ctypes.c_void_p(bb.data_ptr())) == 0
    eng.denoise_step(carr[30], noise[31])
    torch.cuda.synchronize()
    lib.pfk_build_set_stamp_buffer(None)
    st = bb.cpu().view(B, 32)
    print("== k_step_build: cycles [start -> build start (update) -> ff done -> kNN done -> protein side emitted]")
    for g in range(3):
        r = st[g]
        print(f"  graph {g}: update {int(r[8] - r[0])}  ff {int(r[9] - r[8])}  kNN {int(r[10] - r[9])}  emit {int(r[11] - r[10])} (count {int(r[12] - r[10])}, scan {int(r[13] - r[12])}, stores {int(r[14] - r[13])}, barrier {int(r[15] - r[14])}, pa copy {int(r[11] - r[15])})  total {int(r[11] - r[0])}")
OFFS = [int(x) for x in os.environ.get("STAMP_OFFSETS", "0").split(",")]       # first recorded item per pass
lib.pfk_rg_set_stamp_offset.argtypes = [ctypes.c_int]
for which, off in [(w, o) for w in range(4) for o in (OFFS if w == 0 else [0])]:
    assert lib.pfk_rg_set_stamp_offset(off) == 0
    buf.zero_()
    assert lib.pfk_rg_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    lib.pfk_rg_set_stamp_which(which)
    eng.denoise_step(carr[30], noise[31])
    torch.cuda.synchronize()
    lib.pfk_rg_set_stamp_buffer(None)
    st = buf.cpu().view(64, 64)
    print(f"== launch {which}: {names[which]}  items from {off}  (cycles between consecutive stamps; waves that ran)")
    t0 = min(int(st[w][0]) for w in range(64) if int(st[w][0]) != 0) if bool((st[:, 0] != 0).any()) else 0
    shown = 0
    for w in range(64):
        row = st[w]
        n = int((row != 0).sum())
        if n < 2:
            continue
        dl = [int(row[i + 1] - row[i]) for i in range(n - 1)]
        print(f"  wave {w:2d}: start +{int(row[0]) - t0}  total {int(row[n - 1] - row[0])}  " + " ".join(str(x) for x in dl))
        shown += 1
        if shown >= 6:
            break
