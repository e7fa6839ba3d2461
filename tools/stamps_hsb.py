#!/usr/bin/env python3
"""Diagnostic: timeline of the merged last launch of a step (k_rg_node_hs_build) at config 2, s_memrealtime (10 ns ticks, one clock).
    pharmacophore-diffusion_amd/csrc/build_variant.sh hsbst "-DPF_HSB_STAMPS"
    PFDYN_LIB=$PWD/pharmacophore-diffusion_amd/csrc/variants/libpfdyn_hsbst.so python3 tools/stamps_hsb.py"""
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
G = 4096
S = 24
buf = torch.zeros(G * S, dtype=torch.int64, device=dev)
lib = eng.lib
lib.pfk_hsb_set_stamp_buffer.argtypes = [ctypes.c_void_p]
eng.prepare_timesteps(carr)            # (announced timesteps: the launch also computes the next call's center tables and 'pa' rows)
eng.sample_begin(noise[0])
for i in range(30):
    eng.denoise_step(carr[i], noise[i + 1])
torch.cuda.synchronize()
assert eng.kernel_family(2) == 2
assert lib.pfk_hsb_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
eng.denoise_step(carr[30], noise[31])
torch.cuda.synchronize()
lib.pfk_hsb_set_stamp_buffer(None)
st = buf.cpu().view(G, S)
live = [b for b in range(G) if int(st[b, 0]) != 0]
t0 = min(int(st[b, 0]) for b in live)
heads = [b for b in live if b < 192 and int(st[b, 2]) == 0 and int(st[b, 1]) != 0]
ahead = [b for b in live if b >= 224 and int(st[b, 1]) != 0]
builds = [b for b in live if int(st[b, 2]) != 0]
us = lambda v: (int(v) - t0) / 100.0
print(f"{len(live)} workgroups stamped: {len(heads)} node + head, {len(builds)} update + build; times in us from the first workgroup's start")
print("node + head workgroups: start / end")
for b in heads[:6] + heads[-3:]:
    print(f"  wg {b:3d}: {us(st[b, 0]):6.2f} -> {us(st[b, 1]):6.2f}")
print("  last head end:", max(us(st[b, 1]) for b in heads))
print("update + build workgroups: start | loads issued | eps arrived | COM known | coordinates in LDS | ff wave done | wave 0 kNN done | barrier passed | references counted | offsets known | fp stored | slot owners | body end | wg end")
for b in builds[:6] + builds[-3:]:
    print(f"  wg {b:3d}: " + " | ".join(f"{us(st[b, k]):6.2f}" for k in (0, 2, 3, 4, 8, 10, 9, 5, 11, 12, 6, 13, 7, 1)))
import statistics
ks = (3, 4, 8, 10, 9, 5, 11, 12, 6, 13, 7, 1)
print("  median from eps arrived: " + " | ".join(f"{statistics.median(us(st[b, k]) - us(st[b, 3]) for b in builds):6.2f}" for k in ks))
if all(int(st[b, 14]) != 0 for b in builds):     # (only in builds that stamp ids 30.. inside knn_halfwave_keys)
    print("  inside wave 0's neighbour search, from 'coordinates in LDS' (medians): " + " | ".join(f"{statistics.median(us(st[b, k]) - us(st[b, 8]) for b in builds):6.2f}" for k in (14, 15, 16, 17)))
print("  last build end:", max(us(st[b, 1]) for b in builds))
if ahead:
    # slot order: node + head | update + build | speculative "pa" items | center hoist (the hoist's workgroups read 19 or more words: they
    # are the ones whose end comes after eps)
    late = [b for b in ahead if us(st[b, 1]) > min(us(st[b2, 3]) for b2 in builds)]
    early = [b for b in ahead if b not in late]
    for nm, ws in (("ahead workgroups done before eps (speculative pa items)", early), ("ahead workgroups done after eps (center hoist)", late)):
        if ws:
            print(f"  {nm}: {len(ws)}, start {min(us(st[b, 0]) for b in ws):.2f} .. {max(us(st[b, 0]) for b in ws):.2f}, end {min(us(st[b, 1]) for b in ws):.2f} .. {max(us(st[b, 1]) for b in ws):.2f}")
    noend = [b for b in live if b >= 224 and int(st[b, 1]) == 0]
    print(f"  slots >= 224 that started and left without an item: {len(noend)}")
print("what the step computed ahead (pf_debug_ahead):", eng.ahead())
if os.environ.get("HSB_DUMP"):                  # start + duration of every ahead workgroup, by slot
    rows = [(b, us(st[b, 0]), us(st[b, 1])) for b in range(224, G) if int(st[b, 1]) != 0]
    for k in range(0, len(rows), 8):
        print(" ".join(f"{b}:{s_:.1f}+{e_ - s_:.1f}" for b, s_, e_ in rows[k:k + 8]))
