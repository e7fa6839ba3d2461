#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of the n16 edge launches (pf_n16.hip, -DN16_TRACE build): start / end s_memtime, the
compute unit (HW_ID, XCC_ID) and the item kind of every workgroup of one launch.

    cd pharmacophore-diffusion_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 \
        -DN16_TRACE -c pf_n16.hip -o variants/n16_trace.o && hipcc -shared -fPIC --offload-arch=gfx950 pf_kernels.o pf_train.o \
        pf_rg.o variants/n16_trace.o pf_host.o -o variants/libpfdyn_n16trace.so
    PFDYN_N16=2 PFDYN_LIB=$PWD/pharmacophore-diffusion_amd/csrc/variants/libpfdyn_n16trace.so B=32 python tools/n16_trace.py
"""
import collections
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
lib.pfk_n16_set_trace_buffer.argtypes = [ctypes.c_void_p]
eng.sample_begin(noise[0])
for i in range(30):
    eng.denoise_step(carr[i], noise[i + 1])
torch.cuda.synchronize()
NW = 1 << 15
buf = torch.zeros(NW * 4, dtype=torch.int64, device=dev)
assert lib.pfk_n16_set_trace_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
eng.denoise_step(carr[30], noise[31])          # every n16 launch of the step writes the buffer: the last launch wins per block index
torch.cuda.synchronize()
lib.pfk_n16_set_trace_buffer(None)
st = buf.cpu().view(NW, 4).tolist()
rows = [(r[0], r[1], r[2] & 0xffffffff, r[3]) for r in st if r[0] != 0 and r[1] != 0]
print("workgroups recorded:", len(rows))
t0 = min(r[0] for r in rows)
ET = ["ff", "pf", "fp", "pp"]
per_cu = collections.Counter()
for a, b, hw, x in rows:
    cu, sh, se, xcc = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, x & 15
    per_cu[(xcc, se, sh, cu)] += 1
print("compute units used:", len(per_cu), " workgroups per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
durs = collections.defaultdict(list)
for a, b, hw, x in rows:
    durs[ET[(x >> 32) & 3]].append(b - a)
for k, v in durs.items():
    v.sort()
    print(f"  {k}: {len(v)} items, duration ticks min {v[0]} median {v[len(v) // 2]} max {v[-1]}")
ends = sorted(b - t0 for a, b, hw, x in rows)
starts = sorted(a - t0 for a, b, hw, x in rows)
print("start ticks: first", starts[0], "median", starts[len(starts) // 2], "last", starts[-1], "| end ticks: first", ends[0],
      "median", ends[len(ends) // 2], "last", ends[-1])
