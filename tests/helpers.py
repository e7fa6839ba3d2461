"""Shared helpers for the test-suite (CPU side)."""
import os

import numpy as np
import torch

from oracle import pf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" and z[k].ndim > 0 else z[k]) for k in z.files}


def batch_from(z, prefix="b_"):
    return O.PocketBatch(z[prefix + "prot_x"], z[prefix + "prot_h"], z[prefix + "prot_ptr"].long(),
                         z[prefix + "pharm_ptr"].long(), z[prefix + "pp_src"].long(), z[prefix + "pp_dst"].long())


def edge_set(src, dst):
    return set(zip(src.tolist(), dst.tolist()))


# dynamics goldens: file -> config that generated it (tests/golden/make_golden.py:main)
DYN_CASES = {
    "dynamics_c1.npz": O.DynamicsConfig(),
    "dynamics_ragged.npz": O.DynamicsConfig(),
    "dynamics_radius.npz": O.DynamicsConfig(n_convs=3, n_noise_gvps=3, message_norm=10, pf_k=0, ff_k=0),
    "dynamics_knnff.npz": O.DynamicsConfig(ff_k=2, pf_k=3, message_norm=1),
    # message_norm = 0 (per-graph normalisers, gvp.py:504-507): radius pf edges / kNN pf edges (the reference's
    # dynamics_gvp.py:220 bookkeeping, reproduced), ragged pockets
    "dynamics_gnorm_radius.npz": O.DynamicsConfig(message_norm=0, pf_k=0),
    "dynamics_gnorm_knn.npz": O.DynamicsConfig(message_norm=0, pf_k=5),
}


# training goldens (reference in train() mode, with the dropout draws recorded)
GRAD_CASES = {
    "train_grads.npz": O.DynamicsConfig(),
    "train_grads_radius.npz": O.DynamicsConfig(n_convs=3, n_noise_gvps=3, message_norm=10, pf_k=0, ff_k=0),
}


def dropout_from(z, cfg):
    """conv_layer-style mask dicts (one per layer) from a train_grads golden."""
    return [{nt: tuple(z[f"drop_{i}_{nt}_{w}_{c}"] for w in ("msg", "res") for c in ("s", "v"))
             for nt in ("pharm", "prot")} for i in range(cfg.n_convs)]
