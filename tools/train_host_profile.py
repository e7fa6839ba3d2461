#!/usr/bin/env python3
"""Host time of the pieces of a training step (no synchronisation inside the step): where a step is host-bound.
    python tools/train_host_profile.py [batches]   (host enqueue time per step against the device-bound step time, then a cProfile of 100 steps)"""
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pharmacoforge_amd as pfa
from pharmacoforge_amd import synthetic

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
B = 256
dyn = dict(vector_size=16, n_convs=2, n_hidden_scalars=128, message_norm='mean', dropout=0.1, ff_k=0, pf_k=5,
           n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4)
m = pfa.PharmacophoreDiff(6, 11, pfa.analysis.ph_idx_to_type, None, n_timesteps=100,
                          graph_config={'graph_cutoffs': {'pp': 3.5, 'pf': 8, 'fp': 8, 'ff': 9}}, dynamics_config=dyn,
                          precision=1e-5, lr_scheduler_config={'base_lr': 1e-4, 'weight_decay': 1e-12})
sd = dict(synthetic.make_state_dict(0)); sd["gamma.gamma"] = m.state_dict()["gamma.gamma"]
m.load_state_dict(sd, strict=True)
m = m.to(dev).train()
eng = m.dynamics.engine()
pockets = [synthetic.synthetic_pocket(i, 256) for i in range(B)]
gen = torch.Generator().manual_seed(7)
graphs = []
for r in range(nb):
    order = [(i + r * 64) % B for i in range(B)]
    sz = [4 + ((i + r) % 5) for i in range(B)]
    xs, hs = [pockets[i][0] for i in order], [pockets[i][1] for i in order]
    prot_x, prot_h = torch.cat(xs), torch.cat(hs)
    prot_ptr = torch.arange(B + 1, dtype=torch.int64) * 256
    pharm_ptr = torch.tensor([0] + list(itertools.accumulate(sz)), dtype=torch.int64)
    pp_src, pp_dst = eng.build_pp_edges(prot_x.to(dev), prot_ptr)
    Nf = int(pharm_ptr[-1])
    x0 = torch.cat([xs[i].mean(0, keepdim=True) + 2.0 * torch.randn(sz[i], 3, generator=gen) for i in range(B)])
    h0 = torch.nn.functional.one_hot(torch.randint(0, 6, (Nf,), generator=gen), 6).float()
    graphs.append(pfa.PocketGraph(prot_x, prot_h, prot_ptr, pharm_ptr, pp_src, pp_dst, pharm_x0=x0, pharm_h0=h0).to(dev))
opt = pfa.FlatAdam(m.dynamics, lr=1e-4, weight_decay=1e-12)
import cProfile, pstats
def step(it):
    opt.zero_grad(lazy=True)
    g = graphs[it % nb]
    loss = m.training_step(g, 0)
    loss.backward()
    opt.step()
for it in range(20): step(it)
torch.cuda.synchronize()
if "--stalls" in sys.argv:
    # per-step host time over 600 steps (no synchronisation inside): a stall shows as one long step; which call it sits in is
    # narrowed down by the three marks inside the step
    marks = []
    def step_t(it):
        a = time.perf_counter()
        opt.zero_grad(lazy=True)
        g = graphs[it % nb]
        loss = m.training_step(g, 0)
        b = time.perf_counter()
        loss.backward()
        c = time.perf_counter()
        opt.step()
        d = time.perf_counter()
        marks.append((d - a, b - a, c - b, d - c))
    for it in range(600): step_t(it)
    torch.cuda.synchronize()
    worst = sorted(range(len(marks)), key=lambda i: -marks[i][0])[:4]
    med = sorted(x[0] for x in marks)[len(marks) // 2]
    print(f"median step (host) {med * 1e3:.3f} ms; longest: " + "; ".join(
        f"step {i}: {marks[i][0] * 1e3:.2f} ms (forward {marks[i][1] * 1e3:.2f}, backward {marks[i][2] * 1e3:.2f}, optimiser {marks[i][3] * 1e3:.2f})" for i in worst))
    sys.exit(0)
t0 = time.perf_counter()
for it in range(200): step(it)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"200 steps: host enqueue {1e3 * (t1 - t0) / 200:.3f} ms per step, with the device {1e3 * (t2 - t0) / 200:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for it in range(100): step(it)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
