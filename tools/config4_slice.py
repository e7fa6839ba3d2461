#!/usr/bin/env python3
"""Wall-clock of a slice of BASELINE config 4 on one GPU: P synthetic 256-atom pockets x 30 samples (sizes 3,3,3,3,3,4..8
cycling), T=500, max_batch_size 128, through PharmacophoreDiff.sample (graph copies, batching, set_batch, the fused
pf_sample loop, unbatching into SampledPharmacophore objects).
    python tools/config4_slice.py [P] [S] [L]  # S samples per pocket (default 30; 128 = one pocket per batch, the shape of a
                                               # generate_pharmacophores.py run with --samples_per_pocket 128); L batches in
                                               # flight (default: PharmacophoreDiff.sample_lanes)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pharmacoforge_amd as pfa
from pharmacoforge_amd import synthetic

P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = 500
dyn = dict(vector_size=16, n_convs=2, n_hidden_scalars=128, message_norm='mean', dropout=0.1, ff_k=0, pf_k=5,
           n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4)
m = pfa.PharmacophoreDiff(6, 11, pfa.analysis.ph_idx_to_type, None, n_timesteps=T,
                          graph_config={'graph_cutoffs': {'pp': 3.5, 'pf': 8, 'fp': 8, 'ff': 9}}, dynamics_config=dyn, precision=1e-5)
sd = dict(synthetic.make_state_dict(0)); sd["gamma.gamma"] = m.state_dict()["gamma.gamma"]
m.load_state_dict(sd, strict=True)
m = m.to("cuda").eval()
pockets = []
for i in range(P):
    x, h = synthetic.synthetic_pocket(i, 256)
    pockets.append(pfa.build_initial_complex_graph(x, h, cutoffs={'pp': 3.5, 'pf': 8, 'fp': 8, 'ff': 9},
                                                   pharm_atom_positions=torch.zeros(1, 3), pharm_atom_features=torch.zeros(1, 6)))
S = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sizes = (([3] * 5 + [4, 5, 6, 7, 8]) * 13)[:S]
n_pharms = [sizes for _ in range(P)]
LANES = int(sys.argv[3]) if len(sys.argv) > 3 else None
MAXB = int(os.environ.get("MAXB", "128"))          # max_batch_size
torch.manual_seed(0)
with torch.no_grad():
    nw = min(P, max(2, 4 * ((S * 4 + MAXB - 1) // MAXB) // max(S, 1) + 4))       # enough batches to create every lane's handle outside the timed run
    m.sample(pockets[:nw], n_pharms[:nw], max_batch_size=MAXB, lanes=LANES)
    torch.cuda.synchronize()
    t0 = time.time()
    out = m.sample(pockets, n_pharms, max_batch_size=MAXB, lanes=LANES)
    torch.cuda.synchronize()
    dt = time.time() - t0
n = sum(len(o) for o in out)
print(f"[{LANES or m.sample_lanes or 'auto'} lane(s), batches of {MAXB}] {P} pockets x {S} samples = {n} pharmacophores, T={T}: {dt:.2f} s  ->  {n * T / dt / 1e3:.0f} k sample-steps/s end to end, "
      f"{dt / P * 1e3:.0f} ms per pocket")
