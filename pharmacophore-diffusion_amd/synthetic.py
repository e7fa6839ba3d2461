"""Seeded synthetic inputs for benchmarks and smoke tests: pocket
generator, deterministic weights in the reference state-dict key layout, noise schedule-free.
Pure numpy/torch host code; no compute path.  (The CPU oracle keeps its own copy of these
generators; tests/test_host_logic.py checks the two agree bit for bit.)"""
import math
from typing import Dict, List

import numpy as np
import torch

ETYPE_KEYS = ("pharm_ff_pharm", "prot_pf_pharm", "pharm_fp_prot", "prot_pp_prot")


def synthetic_pocket(seed: int, n_prot: int, rec_nf: int = 11):
    """n_prot points uniform in a ball of density 0.05 atoms/A^3, 1.2 A exclusion radius; element
    one-hots with P(C,N,O,S) = (0.62, 0.17, 0.19, 0.02)."""
    rng = np.random.default_rng(seed)
    R = (3.0 * n_prot / (4.0 * math.pi * 0.05)) ** (1.0 / 3.0)
    pts: List[np.ndarray] = []
    while len(pts) < n_prot:
        p = rng.uniform(-R, R, size=3)
        if np.dot(p, p) > R * R:
            continue
        if pts:
            d = np.linalg.norm(np.asarray(pts) - p, axis=1)
            if d.min() < 1.2:
                continue
        pts.append(p)
    x = torch.tensor(np.asarray(pts), dtype=torch.float32)
    el = rng.choice(4, size=n_prot, p=[0.62, 0.17, 0.19, 0.02])
    h = torch.zeros(n_prot, rec_nf, dtype=torch.float32)
    h[torch.arange(n_prot), torch.tensor(el)] = 1.0
    return x, h


def state_dict_spec(pharm_nf=6, rec_nf=11, vector_size=16, n_hidden_scalars=128, rbf_dim=16, n_convs=2,
                    n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4):
    """(name, shape, fan_in) of every tensor under 'dynamics.' in the reference key layout."""
    S, V, R = n_hidden_scalars, vector_size, rbf_dim
    spec = []

    def gvp(p, vi, vo, si, so):
        h = max(vi, vo)
        spec.append((p + "Wh", (vi, h), vi))
        spec.append((p + "Wu", (h, vo), h))
        spec.append((p + "to_feats_out.0.weight", (so, h + si), h + si))
        spec.append((p + "to_feats_out.0.bias", (so,), h + si))
        spec.append((p + "scalar_to_vector_gates.weight", (vo, so), so))
        spec.append((p + "scalar_to_vector_gates.bias", (vo,), so))

    for nt, nf in (("pharm", pharm_nf), ("prot", rec_nf)):
        p = f"dynamics.{nt}_encoder."
        spec += [(p + "0.weight", (S, nf + 1), nf + 1), (p + "0.bias", (S,), nf + 1),
                 (p + "2.weight", (S,), 0), (p + "2.bias", (S,), -1)]
    for i in range(n_convs):
        p = f"dynamics.noise_predictor.conv_layers.{i}."
        for key in ETYPE_KEYS:
            for j in range(n_message_gvps):
                if j == 0:
                    gvp(f"{p}edge_message_fns.{key}.{j}.", V + 1, V, S + R, S)
                else:
                    gvp(f"{p}edge_message_fns.{key}.{j}.", V, V, S, S)
        for nt in ("pharm", "prot"):
            for j in range(n_update_gvps):
                gvp(f"{p}node_update_fns.{nt}.{j}.", V, V, S, S)
        for which in ("update_layer_norms", "message_layer_norms"):
            for nt in ("pharm", "prot"):
                spec += [(f"{p}{which}.{nt}.feat_norm.weight", (S,), 0), (f"{p}{which}.{nt}.feat_norm.bias", (S,), -1)]
        spec.append((f"{p}dropout.vector_dropout.dummy_param", (0,), -2))
    p = "dynamics.noise_predictor.noise_predictor."
    for k in range(n_noise_gvps):
        if k == n_noise_gvps - 1:
            gvp(f"{p}gvps.{k}.", V, 1, S, 64)
        else:
            gvp(f"{p}gvps.{k}.", V, V, S, S)
    spec += [(p + "to_scalar_output.weight", (pharm_nf, 64), 64), (p + "to_scalar_output.bias", (pharm_nf,), 64)]
    return spec


def make_state_dict(seed: int = 0, perturb_norm: bool = True, **arch) -> Dict[str, torch.Tensor]:
    """Deterministic weights (numpy PCG64): uniform +-1/sqrt(fan_in) like the reference initialisers,
    LayerNorm affine parameters perturbed away from (1, 0)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape, fan in state_dict_spec(**arch):
        if fan == -2:
            sd[name] = torch.empty(0)
        elif fan == 0:
            w = np.ones(shape) + (0.1 * rng.standard_normal(shape) if perturb_norm else 0.0)
            sd[name] = torch.tensor(w, dtype=torch.float32)
        elif fan == -1:
            w = 0.1 * rng.standard_normal(shape) if perturb_norm else np.zeros(shape)
            sd[name] = torch.tensor(w, dtype=torch.float32)
        else:
            k = 1.0 / math.sqrt(fan)
            sd[name] = torch.tensor(rng.uniform(-k, k, size=shape), dtype=torch.float32)
    return sd
