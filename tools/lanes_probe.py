#!/usr/bin/env python3
"""Probe: what does dealing ONE batch of B graphs over L concurrent lanes (sub-batches of B / L graphs, each on its own handle
and HIP stream) buy?  Every launch of a small-batch denoising step is latency-bound and leaves most compute units idle, and
graphs are independent (pharmacodiff.py:423-429), so sub-batches can overlap.  Two drivers: whole trajectories through pf_sample
(no host work between steps) and the per-step loop of bench.py's headline (noise drawn per step, pf_denoise_step).

    B=32 T=500 python tools/lanes_probe.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pharmacoforge_amd as pfa  # noqa: E402
from pharmacoforge_amd import schedule, synthetic  # noqa: E402

dev = torch.device('cuda', 0)
B, T = int(os.environ.get("B", "32")), int(os.environ.get("T", "500"))
NP = int(os.environ.get("NP", "256"))
coef = schedule.step_coefficients(schedule.PredefinedNoiseSchedule('polynomial_2', T, float(os.environ.get("PREC", "0.25"))).gamma, T)
sd = synthetic.make_state_dict(0)
pockets = [synthetic.synthetic_pocket(1000 + i, NP) for i in range(B)]


def make_engine(g0, g1):
    eng = pfa.PfEngine(device=dev)
    eng.load_state_dict(sd)
    xs, hs = zip(*pockets[g0:g1])
    px, ph = torch.cat(xs).to(dev), torch.cat(hs).to(dev)
    n = g1 - g0
    pptr = torch.arange(n + 1) * NP
    fptr = torch.arange(n + 1) * 6
    s, d = eng.build_pp_edges(px, pptr)
    eng.set_batch(px, ph, pptr, fptr, s, d)
    return eng


for L in (1, 2, 4, 8):
    if B % L:
        continue
    per = B // L
    engs = [make_engine(k * per, (k + 1) * per) for k in range(L)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(L)]
    arr = engs[0].coef_array(coef, reversed(range(T)))
    bufs = [torch.randn(T + 1, per * 6, 9, device=dev) for _ in range(L)]
    res = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(L):
            with torch.cuda.stream(streams[k]):
                engs[k].sample(arr, T, bufs[k])
        torch.cuda.synchronize()
        res.append(time.perf_counter() - t0)
    dt = sorted(res[1:])[1]
    # the same with one host thread per lane (pf_sample is one C call: ctypes releases the GIL)
    import threading
    res2 = []

    def run(k):
        with torch.cuda.stream(streams[k]):
            engs[k].sample(arr, T, bufs[k])
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(k,)) for k in range(L)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        res2.append(time.perf_counter() - t0)
    dt2 = sorted(res2[1:])[1]
    print(f"lanes {L} x {per} graphs: pf_sample, one host thread per lane {B * T / dt2 / 1e3:8.1f} k sample-steps/s ({dt2 / T * 1e6:.1f} us per step of the {B})", flush=True)
    print(f"lanes {L} x {per} graphs: pf_sample whole trajectory {B * T / dt / 1e3:8.1f} k sample-steps/s ({dt / T * 1e6:.1f} us per step of the {B})", flush=True)
    # per-step driver: noise per step on the lane's stream, pf_denoise_step
    K = 200
    for rep in range(3):
        for k in range(L):
            with torch.cuda.stream(streams[k]):
                engs[k].sample_begin(bufs[k][0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            for k in range(L):
                with torch.cuda.stream(streams[k]):
                    nz = torch.randn(per * 6, 9, device=dev)
                    engs[k].denoise_step(arr[i], nz)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"lanes {L} x {per} graphs: per-step loop             {B * K / dt / 1e3:8.1f} k sample-steps/s ({dt / K * 1e6:.1f} us per step; host enqueue {t_host / K * 1e6:.1f} us)", flush=True)
    del engs
