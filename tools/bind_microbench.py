#!/usr/bin/env python3
"""Where a per-batch bind spends its time: host time of PfEngine.set_batch (asynchronous part) and the device time
behind it, for a training-size batch (256 pockets) and a sampling-size batch (128), rotating through distinct batches.
    python tools/bind_microbench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pharmacoforge_amd as pfa
from pharmacoforge_amd import synthetic

dev = torch.device("cuda", 0)
eng = pfa.PfEngine(device=dev)
eng.load_state_dict(synthetic.make_state_dict(0))
pockets = [synthetic.synthetic_pocket(i, 256) for i in range(256)]


def make(B, shift, on_device):
    order = [(i + shift) % 256 for i in range(B)]
    xs, hs = [pockets[i][0] for i in order], [pockets[i][1] for i in order]
    px, ph = torch.cat(xs), torch.cat(hs)
    pptr = torch.arange(B + 1, dtype=torch.int64) * 256
    fptr = torch.tensor([0] + list(__import__("itertools").accumulate([4 + ((i + shift) % 5) for i in range(B)])), dtype=torch.int64)
    s, d = eng.build_pp_edges(px.to(dev), pptr)
    if on_device:
        px, ph = px.to(dev), ph.to(dev)
    return px, ph, pptr, fptr, s, d


for B, on_device in ((256, True), (128, False)):
    batches = [make(B, 17 * k, on_device) for k in range(4)]
    for rep in range(3):
        for k, b in enumerate(batches):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.set_batch(*b)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            Nf = int(b[3][-1])
            x, hh, t = torch.randn(Nf, 3, device=dev), torch.randn(Nf, 6, device=dev), torch.rand(B, device=dev)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            eng.train_forward(x, hh, t, dropout=0.1, seed=1)
            t4 = time.perf_counter()
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            if True:
                print(f"rep {rep} B={B} batch {k}: set_batch host {1e3 * (t1 - t0):.2f} ms + device tail {1e3 * (t2 - t1):.2f} ms; "
                      f"train_forward host {1e3 * (t4 - t3):.2f} ms + device {1e3 * (t5 - t4):.2f} ms")
