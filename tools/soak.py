#!/usr/bin/env python3
"""Soak run: many rounds of PharmacophoreDiff.sample over changing pockets and batch shapes, then many training steps over
changing batches, watching the device memory the process holds (hipMemGetInfo via torch.cuda.mem_get_info): it must level
off once the largest shapes have been seen -- the handles keep their workspaces across batches and free nothing per call, so a
leak shows as a steady climb.      python tools/soak.py [rounds] [train_steps]"""
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pharmacoforge_amd as pfa
from pharmacoforge_amd import synthetic

R = int(sys.argv[1]) if len(sys.argv) > 1 else 30
TS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
T = 100
dyn = dict(vector_size=16, n_convs=2, n_hidden_scalars=128, message_norm='mean', dropout=0.1, ff_k=0, pf_k=5,
           n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4)
cut = {'pp': 3.5, 'pf': 8, 'fp': 8, 'ff': 9}
m = pfa.PharmacophoreDiff(6, 11, pfa.analysis.ph_idx_to_type, None, n_timesteps=T, graph_config={'graph_cutoffs': cut},
                          dynamics_config=dyn, precision=1e-5)
sd = dict(synthetic.make_state_dict(0)); sd["gamma.gamma"] = m.state_dict()["gamma.gamma"]
m.load_state_dict(sd, strict=True)
m = m.to("cuda").eval()


def used_mib():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


def pocket(seed, n_atoms):
    x, h = synthetic.synthetic_pocket(seed, n_atoms)
    return pfa.build_initial_complex_graph(x, h, cutoffs=cut, pharm_atom_positions=torch.zeros(1, 3), pharm_atom_features=torch.zeros(1, 6))


torch.manual_seed(0)
hist = []
t0 = time.time()
for r in range(R):
    # shapes change every round: 3..10 pockets of 120..300 atoms, 5..40 samples each of 3..8 centers, batches of 16..128
    P = 3 + r % 8
    pockets = [pocket(1000 * r + i, 120 + (37 * (r + i)) % 181) for i in range(P)]
    S = 5 + (7 * r) % 36
    sizes = [[3 + (i + j + r) % 6 for j in range(S)] for i in range(P)]
    with torch.no_grad():
        out = m.sample(pockets, sizes, max_batch_size=(16, 32, 64, 128)[r % 4], lanes=(None, 1, 2, 4)[(r // 4) % 4])
    torch.cuda.synchronize()
    assert sum(len(o) for o in out) == P * S
    assert all(bool(torch.isfinite(ph.ph_coords).all()) for o in out for ph in o)
    hist.append(used_mib())
    if r % 5 == 4 or r == R - 1:
        print(f"sample round {r + 1:3d}: {hist[-1]:9.1f} MiB in use on the device ({time.time() - t0:.0f} s)", flush=True)
late = hist[len(hist) // 2:]
print(f"sampling: max over the second half {max(late):.1f} MiB, min {min(late):.1f} MiB, first round {hist[0]:.1f} MiB")
# the merged last launch of a step hands eps over through polled exchange words (k_rg_node_hs_build): no time-out on any handle, with
# up to four batches in flight on as many streams
engs = [m.dynamics.engine()] + [e[0] for e in m.dynamics.__dict__.get("_lane_engines", {}).values()]
tmo = [e.xchg_timeouts() for e in engs]
fam = [e.kernel_family(2) for e in engs]
print(f"exchange time-outs per handle: {tmo}; form of the last step's end per handle (2 = merged launch): {fam}")
assert sum(tmo) == 0

# ---- training steps over changing batches (a new bind every step, FlatAdam as in bench.py --train)
m.train()
opt = pfa.FlatAdam(m.dynamics, lr=1e-4, weight_decay=1e-12)
eng = m.dynamics.engine()
hist_t = []
t0 = time.time()
for s in range(TS):
    B = 8 + (s * 5) % 57
    xs, hs, sz = [], [], []
    for i in range(B):
        x, h = synthetic.synthetic_pocket(50000 + 64 * (s % 16) + i, 128 + (i * 13 + s) % 129)
        xs.append(x); hs.append(h); sz.append(4 + (i + s) % 5)
    prot_x, prot_h = torch.cat(xs), torch.cat(hs)
    prot_ptr = torch.tensor([0] + list(itertools.accumulate(x.shape[0] for x in xs)), dtype=torch.int64)
    pharm_ptr = torch.tensor([0] + list(itertools.accumulate(sz)), dtype=torch.int64)
    pp_src, pp_dst = eng.build_pp_edges(prot_x.to("cuda"), prot_ptr)
    gen = torch.Generator().manual_seed(s)
    Nf = int(pharm_ptr[-1])
    x0 = torch.cat([xs[i].mean(0, keepdim=True) + 2.0 * torch.randn(sz[i], 3, generator=gen) for i in range(B)])
    h0 = torch.nn.functional.one_hot(torch.randint(0, 6, (Nf,), generator=gen), 6).float()
    g = pfa.PocketGraph(prot_x, prot_h, prot_ptr, pharm_ptr, pp_src, pp_dst, pharm_x0=x0, pharm_h0=h0).to("cuda")
    if s % 40 == 0:                                   # alternate the two precisions of the training step (a new forward each time)
        m.dynamics.set_train_precision("bf16" if (s // 40) % 2 else "f32")
    opt.zero_grad()
    loss = m.training_step(g, 0)
    loss.backward(); opt.step()
    if s % 20 == 19 or s == TS - 1:
        torch.cuda.synchronize()
        hist_t.append(used_mib())
        print(f"train step {s + 1:4d}: loss {float(loss):.4f}  {hist_t[-1]:9.1f} MiB in use ({time.time() - t0:.0f} s)", flush=True)
late = hist_t[len(hist_t) // 2:]
print(f"training: max over the second half {max(late):.1f} MiB, min {min(late):.1f} MiB")
