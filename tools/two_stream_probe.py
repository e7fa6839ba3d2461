#!/usr/bin/env python3
"""Probe: does a config-2 batch finish sooner as S independent sub-batches on S HIP streams (one handle each)?
Every launch of a B=32 step has fewer row groups than the chip has SIMDs, so concurrent sub-batches overlap.
    python tools/two_stream_probe.py              # one batch of B split over S streams
    FULL=1 B=128 python tools/two_stream_probe.py # S whole batches of B in flight at once (S handles, S streams)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pharmacoforge_amd as pfa
from pharmacoforge_amd import schedule, synthetic

dev = torch.device("cuda", 0)
B, T = int(os.environ.get("B", "32")), 500
FULL = os.environ.get("FULL", "0") != "0"
pockets = [synthetic.synthetic_pocket(1000 + i, 256) for i in range(B * (4 if FULL else 1))]
coef = schedule.step_coefficients(schedule.PredefinedNoiseSchedule('polynomial_2', T, 1e-5).gamma, T)
sd = synthetic.make_state_dict(0)


def run(S):
    engs, streams, noises = [], [], []
    per = B if FULL else B // S
    for k in range(S):
        eng = pfa.PfEngine(device=dev)
        eng.load_state_dict(sd)
        xs = torch.cat([p[0] for p in pockets[k * per:(k + 1) * per]])
        hs = torch.cat([p[1] for p in pockets[k * per:(k + 1) * per]])
        pptr = torch.arange(per + 1) * 256
        fptr = torch.arange(per + 1) * 6
        s, d = eng.build_pp_edges(xs.to(dev), pptr)
        eng.set_batch(xs, hs, pptr, fptr, s, d)
        engs.append(eng)
        streams.append(torch.cuda.Stream(device=dev))
        noises.append(torch.randn(T + 1, per * 6, 9, device=dev))
    arr = engs[0].coef_array(coef, reversed(range(T)))
    torch.cuda.synchronize()
    times = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                outs.append(engs[k].sample(arr, T, noises[k]))
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = sorted(times[1:])[1]
    tot = per * S
    print(f"{tot} graphs as {S} batch(es) of {per} on {S} stream(s): {best * 1e3:.2f} ms for T={T} -> {tot * T / best / 1e3:.0f} k sample-steps/s", flush=True)


for S in (1, 2, 4):
    run(S)
