#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of k_step_build_fast's phases (pf_stepbuild.h: SB_STAMP / SB_PHASE ids) at config 2.
    pharmacophore-diffusion_amd/csrc/build_variant.sh stamps "-DPF_STAMPS"
    PFDYN_HS_BUILD=0 PFDYN_LIB=$PWD/pharmacophore-diffusion_amd/csrc/variants/libpfdyn_stamps.so python3 tools/stamps_build.py
ids: 0 body start | 8 COM known | 9 coordinates shifted, ff emitted (last wave) | 10 kNN done | 12 references counted | 13 offsets
known | 14 fp edges / descriptors stored | 15 owners staged | 11 end of the body.  Ticks of s_memtime (100 MHz on gfx950: 10 ns)."""
import ctypes, os, sys
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
buf = torch.zeros(64 * 32, dtype=torch.int64, device=dev)
lib = eng.lib
lib.pfk_build_set_stamp_buffer.argtypes = [ctypes.c_void_p]
eng.sample_begin(noise[0])
for i in range(30):
    eng.denoise_step(carr[i], noise[i + 1])
torch.cuda.synchronize()
assert lib.pfk_build_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
eng.denoise_step(carr[30], noise[31])
torch.cuda.synchronize()
lib.pfk_build_set_stamp_buffer(None)
st = buf.cpu().view(64, 32)
order = [0, 8, 9, 10, 12, 13, 14, 15, 11]
t0 = int(st[:B, 0].min())
for g in (0, 1, 7, 15, 31):
    row = st[g]
    print(f"graph {g}: start +{int(row[0]) - t0} | " + " ".join(f"{a}->{b}:{int(row[b] - row[a])}" for a, b in zip(order[:-1], order[1:])) +
          f" | body {int(row[11] - row[0])} ticks")
print("all graphs: last end - first start =", int(st[:B, 11].max()) - t0, "ticks")
