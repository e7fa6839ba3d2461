#!/usr/bin/env python3
"""Diagnostic: start and end of every workgroup of one k_n16_edge_u launch (config 2) on the device-wide 100 MHz clock.
    pharmacophore-diffusion_amd/csrc/build_variant.sh n16stamps "-DN16_STAMPS -DN16_STAMPS_SPARSE"
    PFDYN_LIB=$PWD/pharmacophore-diffusion_amd/csrc/variants/libpfdyn_n16stamps.so python3 tools/edge_u_times.py"""
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
coef = schedule.step_coefficients(schedule.PredefinedNoiseSchedule('polynomial_2', T, float(os.environ.get("PREC", "1e-5"))).gamma, T)
carr = eng.coef_array(coef, list(range(39, -1, -1)))
noise = torch.randn(41, B * 6, 9, device=dev)
lib = eng.lib
lib.pfk_n16_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.pfk_n16_set_stamp_kernel.argtypes = [ctypes.c_int]
eng.prepare_timesteps(carr)
eng.sample_begin(noise[0])
for i in range(30):
    eng.denoise_step(carr[i], noise[i + 1])
torch.cuda.synchronize()
buf = torch.zeros(64 * 4 * 64, dtype=torch.int64, device=dev)
KID = int(os.environ.get("KID", "100"))                  # 100: k_n16_edge_u, 101: k_n16_fused_u
lib.pfk_n16_set_stamp_kernel(KID)
BACK = int(os.environ.get("BACK", "0"))                 # > 0: that many steps back to back (noise drawn per step, as bench.py does) in front of the stamped one, no host wait in between
for step in range(30, 30 + int(os.environ.get("STEPS", "4"))):
    buf.zero_()
    torch.cuda.synchronize()
    assert lib.pfk_n16_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()), 0) == 0
    if BACK:
        eng.sample_begin(noise[0])
        for i in range(BACK):
            nz = torch.randn(B * 6, 9, device=dev)
            eng.denoise_step(carr[i], nz)                   # (every step stamps; the last one's stamps remain)
        nz = torch.randn(B * 6, 9, device=dev)
        eng.denoise_step(carr[BACK], nz)
    else:
        eng.denoise_step(carr[step], noise[step + 1])
    torch.cuda.synchronize()
    lib.pfk_n16_set_stamp_buffer(None, 0)
    st = buf.cpu().view(-1, 2)
    live = [b for b in range(st.shape[0]) if int(st[b, 0]) and int(st[b, 1])]
    t0 = min(int(st[b, 0]) for b in live)
    work = [b for b in live if int(st[b, 1]) - int(st[b, 0]) > 200]           # (> 2 us: it ran an item)
    per_x = {x: [b for b in work if b % 8 == x] for x in range(8)}
    print(f"step {step}: {len(live)} workgroups, {len(work)} with an item, last end {max((int(st[b, 1]) - t0) / 100 for b in live):.2f} us; per XCD (slot & 7): first start / median duration / last end")
    import statistics
    print("   " + " | ".join(f"x{x}: {min((int(st[b, 0]) - t0) / 100 for b in ws):.2f} / {statistics.median((int(st[b, 1]) - int(st[b, 0])) / 100 for b in ws):.2f} / {max((int(st[b, 1]) - t0) / 100 for b in ws):.2f}" for x, ws in per_x.items() if ws))
    if os.environ.get("DUMP"):
        for k in range(0, len(work), 8):
            print("  " + " ".join(f"{b}:{(int(st[b, 0]) - t0) / 100:.2f}+{(int(st[b, 1]) - int(st[b, 0])) / 100:.2f}" for b in work[k:k + 8]))
