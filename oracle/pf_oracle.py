"""CPU oracle for the PharmacoForge denoising hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 CPU restatement (DGL-free, torch_cluster-free) of the
reference algorithm.  It is *not* product code: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker / baseline.
The product path (``pharmacophore-diffusion_amd``) never imports anything from ``oracle/``.

Pinning status (see DESIGN.md "Oracle"): the reference ships no tests or golden vectors
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference code itself,
generated in the build container by ``tests/golden/make_golden.py``:
  * GVP / GVPLayerNorm / _rbf / NoisePredictionBlock / PredefinedNoiseSchedule /
    sigma / alpha / sigma_and_alpha_t_given_s run natively from /root/reference (pure torch);
  * GVPMultiEdgeConv.forward, PharmRecDynamicsGVP.forward, sample_p_zs_given_zt,
    sample_given_receptor and PharmacophoreDiff.forward run from /root/reference on top of
    a functional shim of the DGL / torch_cluster entry points they touch (those libraries
    are absent from the image); that part is pinned "through the shim", i.e. up to the
    documented third-party semantics listed in SURVEY.md section 8(c).

Every function cites the reference file:line (relative to /root/reference) it follows.

Data model (DGL-free):  a batch of B pocket graphs is a ``PocketBatch`` of flat tensors;
node ids are local to their node type ('prot' / 'pharm'), graphs are contiguous ranges
given by ``prot_ptr`` / ``pharm_ptr`` (CSR-style, length B+1).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

ETYPES = ("ff", "pf", "fp", "pp")
# (src ntype, etype, dst ntype)  -- dynamics_gvp.py:46-54
CANONICAL = {
    "ff": ("pharm", "ff", "pharm"),
    "pf": ("prot", "pf", "pharm"),
    "fp": ("pharm", "fp", "prot"),
    "pp": ("prot", "pp", "prot"),
}


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class DynamicsConfig:
    """Hyper-parameters of PharmRecDynamicsGVP (dynamics_gvp.py:96-97) + graph cutoffs.

    Defaults are configs/dev.yml:67-90 (the only shipped config)."""
    pharm_nf: int = 6
    rec_nf: int = 11
    vector_size: int = 16
    n_convs: int = 2
    n_hidden_scalars: int = 128
    message_norm: object = "mean"
    n_message_gvps: int = 3
    n_update_gvps: int = 2
    n_noise_gvps: int = 4
    ff_k: int = 0
    pf_k: int = 5
    cutoff_pp: float = 3.5
    cutoff_pf: float = 8.0
    cutoff_fp: float = 8.0
    cutoff_ff: float = 9.0
    rbf_dmax: float = 15.0   # gvp.py:350 (PharmRecGVP never overrides it)
    rbf_dim: int = 16        # gvp.py:350


@dataclass
class PocketBatch:
    prot_x: torch.Tensor          # [Np_tot, 3] fp32
    prot_h: torch.Tensor          # [Np_tot, rec_nf] fp32
    prot_ptr: torch.Tensor        # [B+1] int64
    pharm_ptr: torch.Tensor       # [B+1] int64
    pp_src: torch.Tensor          # [Epp] int64 (prot-local ids over the whole batch)
    pp_dst: torch.Tensor          # [Epp] int64

    @property
    def batch_size(self) -> int:
        return int(self.prot_ptr.numel() - 1)

    def batch_idxs(self) -> Dict[str, torch.Tensor]:
        """utils/unorganized_utils.py:83-95 get_batch_idxs (repeat_interleave of graph ids)."""
        B = self.batch_size
        ar = torch.arange(B)
        return {
            "prot": ar.repeat_interleave(self.prot_ptr[1:] - self.prot_ptr[:-1]),
            "pharm": ar.repeat_interleave(self.pharm_ptr[1:] - self.pharm_ptr[:-1]),
        }


# --------------------------------------------------------------------------------------
# primitives (gvp.py)
# --------------------------------------------------------------------------------------
def norm_no_nan(x, axis=-1, keepdims=False, eps=1e-8, sqrt=True):
    """gvp.py:12-19."""
    out = torch.clamp(torch.sum(torch.square(x), axis, keepdims), min=eps)
    return torch.sqrt(out) if sqrt else out


def rbf(D, D_min=0.0, D_max=20.0, D_count=16):
    """gvp.py:26-41."""
    D_mu = torch.linspace(D_min, D_max, D_count).view(1, -1)
    D_sigma = (D_max - D_min) / D_count
    return torch.exp(-((D.unsqueeze(-1) - D_mu) / D_sigma) ** 2)


def gvp_forward(sd: Dict[str, torch.Tensor], prefix: str, feats, vectors, vec_act: str = "sigmoid"):
    """GVP.forward, gvp.py:89-116 (vector_gating=True, feats_activation=SiLU).

    ``prefix`` addresses Wh, Wu, to_feats_out.0.{weight,bias}, scalar_to_vector_gates.{weight,bias}
    in the reference state-dict key layout."""
    Wh = sd[prefix + "Wh"]
    Wu = sd[prefix + "Wu"]
    Vh = torch.einsum("bvc,vh->bhc", vectors.float(), Wh)
    Vu = torch.einsum("bhc,hu->buc", Vh, Wu)
    sh = norm_no_nan(Vh)
    s = torch.cat((feats.float(), sh), dim=1)
    feats_out = F.silu(F.linear(s, sd[prefix + "to_feats_out.0.weight"], sd[prefix + "to_feats_out.0.bias"]))
    gating = F.linear(feats_out, sd[prefix + "scalar_to_vector_gates.weight"],
                      sd[prefix + "scalar_to_vector_gates.bias"]).unsqueeze(-1)
    if vec_act == "sigmoid":
        gating = torch.sigmoid(gating)
    elif vec_act != "identity":
        raise ValueError(vec_act)
    return feats_out, gating * Vu


def gvp_chain(sd, prefix: str, n: int, feats, vectors, last_identity: bool = False):
    """nn.Sequential of GVPs: keys ``{prefix}{i}.`` (gvp.py:415,433; dynamics_gvp.py:33)."""
    for i in range(n):
        act = "identity" if (last_identity and i == n - 1) else "sigmoid"
        feats, vectors = gvp_forward(sd, f"{prefix}{i}.", feats, vectors, act)
    return feats, vectors


def gvp_layernorm(sd, prefix: str, feats, vectors, eps: float = 1e-5):
    """GVPLayerNorm.forward, gvp.py:159-166."""
    normed_feats = F.layer_norm(feats, (feats.shape[-1],), sd[prefix + "feat_norm.weight"],
                                sd[prefix + "feat_norm.bias"], 1e-5)
    vn = norm_no_nan(vectors, axis=-1, keepdims=True, sqrt=False)
    vn = torch.sqrt(torch.mean(vn, dim=-2, keepdim=True) + eps) + eps
    return normed_feats, vectors / vn


# --------------------------------------------------------------------------------------
# neighbour search (torch_cluster semantics; SURVEY.md 8(c) "unpinned corners" are fixed here)
# --------------------------------------------------------------------------------------
def _d2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Squared distances [len(a), len(b)] in fp32, evaluated as (dx*dx + dy*dy) + dz*dz
    with one rounding per operation (no FMA) -- the HIP kernels use the same order so the
    edge sets are bit-identical."""
    d = a[:, None, :] - b[None, :, :]
    sq = d * d
    return (sq[..., 0] + sq[..., 1]) + sq[..., 2]


def radius(x, y, r: float, ptr_x, ptr_y, max_num_neighbors: int):
    """torch_cluster.radius(x, y, r, batch_x, batch_y, max_num_neighbors): for every y_i the
    x_j of the same graph with |x_j - y_i|^2 < r^2 (strict), first ``max_num_neighbors`` in
    ascending j.  Returns [2, E]: row 0 = y index, row 1 = x index, grouped by ascending i."""
    rows, cols = [], []
    r2 = torch.tensor(r, dtype=torch.float32) ** 2
    for g in range(len(ptr_x) - 1):
        x0, x1 = int(ptr_x[g]), int(ptr_x[g + 1])
        y0, y1 = int(ptr_y[g]), int(ptr_y[g + 1])
        if x1 == x0 or y1 == y0:
            continue
        d2 = _d2(y[y0:y1], x[x0:x1])
        mask = d2 < r2
        # keep first max_num_neighbors per row
        keep = mask & (torch.cumsum(mask.to(torch.int64), dim=1) <= max_num_neighbors)
        yi, xj = torch.nonzero(keep, as_tuple=True)
        rows.append(yi + y0)
        cols.append(xj + x0)
    if not rows:
        return torch.zeros((2, 0), dtype=torch.int64)
    return torch.stack([torch.cat(rows), torch.cat(cols)])


def radius_graph(x, r: float, ptr, max_num_neighbors: int):
    """torch_cluster.radius_graph(x, r, batch, max_num_neighbors), flow source_to_target, no
    self loops.  Returns [2, E]: row 0 = source j, row 1 = target i, grouped by ascending i.
    (dynamics_gvp.py:196, dataset/protein_pharm_dataset.py:235)."""
    rows, cols = [], []
    r2 = torch.tensor(r, dtype=torch.float32) ** 2
    for g in range(len(ptr) - 1):
        a, b = int(ptr[g]), int(ptr[g + 1])
        if b - a < 2:
            continue
        d2 = _d2(x[a:b], x[a:b])
        mask = d2 < r2
        mask.fill_diagonal_(False)
        keep = mask & (torch.cumsum(mask.to(torch.int64), dim=1) <= max_num_neighbors)
        ti, sj = torch.nonzero(keep, as_tuple=True)
        rows.append(sj + a)
        cols.append(ti + a)
    if not rows:
        return torch.zeros((2, 0), dtype=torch.int64)
    return torch.stack([torch.cat(rows), torch.cat(cols)])


def knn(x, y, k: int, ptr_x, ptr_y):
    """torch_cluster.knn(x, y, k, batch_x, batch_y): for every y_i the min(k, n_x) nearest x_j
    of the same graph, ordered by (d^2, j) ascending.  Returns [2, E]: row 0 = y index,
    row 1 = x index (dynamics_gvp.py:202)."""
    rows, cols = [], []
    for g in range(len(ptr_x) - 1):
        x0, x1 = int(ptr_x[g]), int(ptr_x[g + 1])
        y0, y1 = int(ptr_y[g]), int(ptr_y[g + 1])
        if x1 == x0 or y1 == y0:
            continue
        d2 = _d2(y[y0:y1], x[x0:x1])
        kk = min(k, x1 - x0)
        order = torch.argsort(d2, dim=1, stable=True)[:, :kk]   # stable => ties by ascending j
        yi = torch.arange(y1 - y0).repeat_interleave(kk)
        rows.append(yi + y0)
        cols.append(order.reshape(-1) + x0)
    if not rows:
        return torch.zeros((2, 0), dtype=torch.int64)
    return torch.stack([torch.cat(rows), torch.cat(cols)])


def knn_graph(x, k: int, ptr):
    """torch_cluster.knn_graph(x, k, batch), loop=False, flow source_to_target
    (dynamics_gvp.py:194).  Row 0 = source (neighbour) j, row 1 = target i."""
    rows, cols = [], []
    for g in range(len(ptr) - 1):
        a, b = int(ptr[g]), int(ptr[g + 1])
        n = b - a
        if n < 2:
            continue
        d2 = _d2(x[a:b], x[a:b])
        d2 = d2.clone()
        d2.fill_diagonal_(float("inf"))
        kk = min(k, n - 1)
        order = torch.argsort(d2, dim=1, stable=True)[:, :kk]
        ti = torch.arange(n).repeat_interleave(kk)
        rows.append(order.reshape(-1) + a)
        cols.append(ti + a)
    if not rows:
        return torch.zeros((2, 0), dtype=torch.int64)
    return torch.stack([torch.cat(rows), torch.cat(cols)])


def build_pp_edges(prot_x, prot_ptr, cutoff: float = 3.5, max_num_neighbors: int = 100):
    """Static prot-prot edges: dataset/protein_pharm_dataset.py:234-236 applied per graph."""
    e = radius_graph(prot_x, cutoff, prot_ptr, max_num_neighbors)
    return e[0], e[1]


def build_dynamic_edges(cfg: DynamicsConfig, batch: PocketBatch, prot_x, pharm_x):
    """add_pharm_edges, dynamics_gvp.py:187-215.  Returns {etype: (src, dst)} (local ids)."""
    edges = {}
    if cfg.ff_k > 0:
        ff = knn_graph(pharm_x, cfg.ff_k, batch.pharm_ptr)
    else:
        ff = radius_graph(pharm_x, cfg.cutoff_ff, batch.pharm_ptr, 200)
    edges["ff"] = (ff[0], ff[1])
    if cfg.pf_k > 0:
        pf = knn(prot_x, pharm_x, cfg.pf_k, batch.prot_ptr, batch.pharm_ptr)
        # row0 = pharm (query y), row1 = prot (x);  pf: src=prot, dst=pharm (:206); fp reversed (:209)
        edges["pf"] = (pf[1], pf[0])
        edges["fp"] = (pf[0], pf[1])
    else:
        pf = radius(pharm_x, prot_x, cfg.cutoff_pf, batch.pharm_ptr, batch.prot_ptr, 100)
        # row0 = prot (query y), row1 = pharm (x);  pf: src=row0=prot, dst=row1=pharm (:212)
        edges["pf"] = (pf[0], pf[1])
        edges["fp"] = (pf[1], pf[0])
    return edges


# --------------------------------------------------------------------------------------
# conv layer / dynamics (gvp.py:459-551, dynamics_gvp.py:84-185)
# --------------------------------------------------------------------------------------
def _aggregate(msg, dst, n_dst: int, mean: bool):
    """copy_e + sum|mean reducer: zero rows for in-degree-0 nodes (DGL semantics)."""
    out = torch.zeros((n_dst,) + tuple(msg.shape[1:]), dtype=msg.dtype)
    out.index_add_(0, dst, msg)
    if mean:
        deg = torch.zeros(n_dst, dtype=msg.dtype)
        deg.index_add_(0, dst, torch.ones(dst.shape[0], dtype=msg.dtype))
        deg = deg.clamp(min=1.0)
        out = out / deg.view(-1, *([1] * (msg.dim() - 1)))
    return out


def conv_layer(sd, prefix: str, cfg: DynamicsConfig, node_feats, edges, batch: PocketBatch,
               edge_counts=None, dropout=None):
    """GVPMultiEdgeConv.forward, gvp.py:459-538.

    node_feats: {ntype: (h [N,S], x [N,3], v [N,V,3])};  edges: {etype: (src, dst)}.
    dropout: None (eval mode: identity) or {ntype: (msg_s [N,S], msg_v [N,V], res_s [N,S],
    res_v [N,V])} multiplicative masks with entries in {0, 1/(1-p)} -- the draws of GVPDropout
    (gvp.py:118-149) at its two call sites (gvp.py:518,529), injected."""
    mean = cfg.message_norm == "mean"
    agg_s = {nt: None for nt in ("pharm", "prot")}
    agg_v = {nt: None for nt in ("pharm", "prot")}
    for et in ETYPES:                                   # etypes order: dynamics_gvp.py:46-54
        s_nt, _, d_nt = CANONICAL[et]
        src, dst = edges[et]
        hs, xs, vs = node_feats[s_nt]
        _, xd, _ = node_feats[d_nt]
        n_dst = xd.shape[0]
        x_diff = xs[src] - xd[dst]                       # fn.u_sub_v, gvp.py:474
        dij = norm_no_nan(x_diff, keepdims=True) + 1e-8  # gvp.py:478
        x_diff = x_diff / dij                            # gvp.py:479
        d = rbf(dij.squeeze(1), D_max=cfg.rbf_dmax, D_count=cfg.rbf_dim)   # gvp.py:480
        vec_feats = torch.cat([x_diff.unsqueeze(1), vs[src]], dim=1)       # gvp.py:545
        scalar_feats = torch.cat([hs[src], d], dim=1)                      # gvp.py:547
        key = "_".join(CANONICAL[et])
        ms, mv = gvp_chain(sd, f"{prefix}edge_message_fns.{key}.", cfg.n_message_gvps,
                           scalar_feats, vec_feats)                        # gvp.py:549
        a_s = _aggregate(ms, dst, n_dst, mean)           # gvp.py:488-497
        a_v = _aggregate(mv, dst, n_dst, mean)
        agg_s[d_nt] = a_s if agg_s[d_nt] is None else agg_s[d_nt] + a_s    # cross_reducer 'sum'
        agg_v[d_nt] = a_v if agg_v[d_nt] is None else agg_v[d_nt] + a_v
    out = {}
    bidx = batch.batch_idxs()
    for nt in ("pharm", "prot"):
        h, x, v = node_feats[nt]
        if mean:
            norm_value = 1.0                               # gvp.py:380-381
        elif cfg.message_norm == 0:
            # gvp.py:504-507: per-graph (edges into ntype)/(nodes of ntype) + 1
            assert edge_counts is not None
            ptr = batch.pharm_ptr if nt == "pharm" else batch.prot_ptr
            n_nodes = (ptr[1:] - ptr[:-1]).to(torch.float32)
            tot = sum(edge_counts[et] for et in ETYPES if CANONICAL[et][2] == nt).to(torch.float32)
            norm_value = (tot / n_nodes + 1)[bidx[nt]].unsqueeze(1)
        else:
            norm_value = cfg.message_norm
        sm = agg_s[nt] / norm_value                         # gvp.py:512
        nv = norm_value.unsqueeze(-1) if isinstance(norm_value, torch.Tensor) else norm_value
        vm = agg_v[nt] / nv                                 # gvp.py:517
        if dropout is not None:                             # gvp.py:518
            sm = sm * dropout[nt][0]
            vm = vm * dropout[nt][1].unsqueeze(-1)
        h1 = h + sm
        v1 = v + vm
        h1, v1 = gvp_layernorm(sd, f"{prefix}message_layer_norms.{nt}.", h1, v1)   # gvp.py:521
        rs, rv = gvp_chain(sd, f"{prefix}node_update_fns.{nt}.", cfg.n_update_gvps, h1, v1)  # :524
        if dropout is not None:                             # gvp.py:529
            rs = rs * dropout[nt][2]
            rv = rv * dropout[nt][3].unsqueeze(-1)
        h2 = h1 + rs
        v2 = v1 + rv
        h2, v2 = gvp_layernorm(sd, f"{prefix}update_layer_norms.{nt}.", h2, v2)    # gvp.py:532
        out[nt] = (h2, x, v2)
    return out


def encode(sd, prefix: str, h, t_node):
    """Linear + SiLU + LayerNorm on [h, t], dynamics_gvp.py:107-117,143-151."""
    s = torch.cat([h, t_node.view(-1, 1)], dim=1)
    s = F.silu(F.linear(s, sd[prefix + "0.weight"], sd[prefix + "0.bias"]))
    return F.layer_norm(s, (s.shape[-1],), sd[prefix + "2.weight"], sd[prefix + "2.bias"], 1e-5)


def noise_head(sd, prefix: str, cfg: DynamicsConfig, h, v):
    """NoisePredictionBlock.forward, dynamics_gvp.py:37-42."""
    s, vec = gvp_chain(sd, prefix + "gvps.", cfg.n_noise_gvps, h, v, last_identity=True)
    s = F.linear(s, sd[prefix + "to_scalar_output.weight"], sd[prefix + "to_scalar_output.bias"])
    return s, vec.squeeze(1)


def _edges_per_graph(node_idx, ptr):
    """get_edges_per_batch, utils/unorganized_utils.py:17-23: ``node_batch_idxs[edge_node_idxs]`` run-length
    encoded with unique_consecutive and scattered into a [B] vector (a graph that appears in two separate runs
    keeps the LAST run's count -- plain assignment, :22; with the index rows the reference passes the runs are
    already grouped by graph)."""
    B = ptr.numel() - 1
    gid = torch.searchsorted(ptr[1:].contiguous(), node_idx, right=True)
    runs, counts = torch.unique_consecutive(gid, return_counts=True)
    out = torch.zeros(B, dtype=torch.int64)
    out[runs] = counts
    return out


def dynamic_edge_counts(cfg: DynamicsConfig, batch: PocketBatch, edges):
    """The per-graph edge counts add_pharm_edges stores on the graph (dynamics_gvp.py:218-225), which gvp.py:506
    reads back when message_norm == 0.

    ff: by the graph of the edge's SOURCE center (``ff_idxs[0]`` with the pharm batch vector, :219) -- source and
    target share a graph, so this is the true count.
    pf: ``get_edges_per_batch(pf_idxs[0], batch_size, prot_batch_idx)`` (:220).  In the radius branch ``pf_idxs[0]``
    holds PROTEIN indices (row 0 of torch_cluster.radius(x=pharm, y=prot) is the y index, :211): the true count.  In
    the kNN branch ``pf_idxs[0]`` holds PHARMACOPHORE-center indices (row 0 of knn(x=prot, y=pharm), :202), which
    the reference nevertheless looks up in the PROTEIN batch vector: center j is booked on the graph that owns
    protein atom j.  That is what the reference computes, so it is what is reproduced here (center indices are
    < Nf_tot <= Np_tot in every usable batch; the reference would raise an IndexError otherwise).
    fp: the same vector as pf (:221).  pp: the static counts of the batched graph."""
    ff_src = edges["ff"][0]
    if cfg.pf_k > 0:
        pf_row0 = edges["pf"][1]          # knn row 0 = pharm (y) index; pf was added as (row1 -> row0), :206
        if pf_row0.numel() and int(pf_row0.max()) >= int(batch.prot_ptr[-1]):
            raise IndexError("dynamics_gvp.py:220 indexes the protein batch vector with a pharmacophore index "
                             "beyond the number of protein atoms")
    else:
        pf_row0 = edges["pf"][0]          # radius row 0 = prot (y) index; pf was added as (row0 -> row1), :212
    pf_cnt = _edges_per_graph(pf_row0, batch.prot_ptr)
    return {
        "ff": _edges_per_graph(ff_src, batch.pharm_ptr),
        "pf": pf_cnt,
        "fp": pf_cnt,
        "pp": _edges_per_graph(edges["pp"][1], batch.prot_ptr),
    }


def dynamics_forward(sd, cfg: DynamicsConfig, batch: PocketBatch, prot_x, pharm_x, pharm_h, t,
                     prefix: str = "dynamics.", return_edges: bool = False, dropout=None):
    """PharmRecDynamicsGVP.forward, dynamics_gvp.py:131-185.

    prot_x: current (COM-shifted) protein coordinates [Np,3]; pharm_x/pharm_h: x_t, h_t;
    t: [B] fp32.  Returns (eps_h [Nf,pharm_nf], eps_x [Nf,3]).  dropout: None or one
    conv_layer mask dict per layer (training mode)."""
    bidx = batch.batch_idxs()
    hp = encode(sd, prefix + "pharm_encoder.", pharm_h, t[bidx["pharm"]])
    hr = encode(sd, prefix + "prot_encoder.", batch.prot_h, t[bidx["prot"]])
    V = cfg.vector_size
    node = {
        "pharm": (hp, pharm_x, torch.zeros(hp.shape[0], V, 3)),
        "prot": (hr, prot_x, torch.zeros(hr.shape[0], V, 3)),
    }
    edges = build_dynamic_edges(cfg, batch, prot_x, pharm_x)
    edges["pp"] = (batch.pp_src, batch.pp_dst)
    edge_counts = None
    if cfg.message_norm == 0 and cfg.message_norm != "mean":
        # dynamics_gvp.py:218-225, including the kNN branch's lookup of center indices in the protein batch
        # vector (:220), reproduced as the reference computes it
        edge_counts = dynamic_edge_counts(cfg, batch, edges)
    for i in range(cfg.n_convs):
        node = conv_layer(sd, f"{prefix}noise_predictor.conv_layers.{i}.", cfg, node, edges, batch,
                          edge_counts, None if dropout is None else dropout[i])
    hp, _, vp = node["pharm"]
    eps_h, eps_x = noise_head(sd, prefix + "noise_predictor.noise_predictor.", cfg, hp, vp)
    if return_edges:
        return eps_h, eps_x, edges
    return eps_h, eps_x


# --------------------------------------------------------------------------------------
# noise schedule (pharmacodiff.py:602-668) and step algebra (pharmacodiff.py:140-160,380-431)
# --------------------------------------------------------------------------------------
def clip_noise_schedule(alphas2, clip_value=0.001):
    """pharmacodiff.py:602-615."""
    alphas2 = np.concatenate([np.ones(1), alphas2], axis=0)
    alphas_step = alphas2[1:] / alphas2[:-1]
    alphas_step = np.clip(alphas_step, a_min=clip_value, a_max=1.0)
    return np.cumprod(alphas_step, axis=0)


def polynomial_schedule(timesteps: int, s=1e-4, power=3.0):
    """pharmacodiff.py:618-632."""
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    alphas2 = (1 - np.power(x / steps, power)) ** 2
    alphas2 = clip_noise_schedule(alphas2, clip_value=0.001)
    precision = 1 - 2 * s
    return precision * alphas2 + s


def gamma_table(timesteps: int, precision: float, power: float = 2.0) -> torch.Tensor:
    """PredefinedNoiseSchedule('polynomial_2'), pharmacodiff.py:641-664 -> fp32 [T+1]."""
    alphas2 = polynomial_schedule(timesteps, s=precision, power=power)
    sigmas2 = 1 - alphas2
    return torch.from_numpy(-(np.log(alphas2) - np.log(sigmas2))).float()


def gamma_lookup(gamma: torch.Tensor, t: torch.Tensor, timesteps: int):
    """PredefinedNoiseSchedule.forward, pharmacodiff.py:666-668."""
    return gamma[torch.round(t * timesteps).long()]


def sigma(gamma):
    return torch.sqrt(torch.sigmoid(gamma))       # pharmacodiff.py:140-142


def alpha(gamma):
    return torch.sqrt(torch.sigmoid(-gamma))      # pharmacodiff.py:144-146


def sigma_and_alpha_t_given_s(gamma_t, gamma_s):
    """pharmacodiff.py:148-160."""
    sigma2_t_given_s = -torch.expm1(F.softplus(gamma_s) - F.softplus(gamma_t))
    log_alpha2_t = F.logsigmoid(-gamma_t)
    log_alpha2_s = F.logsigmoid(-gamma_s)
    alpha_t_given_s = torch.exp(0.5 * (log_alpha2_t - log_alpha2_s))
    alpha_s = torch.exp(0.5 * log_alpha2_s)
    return sigma2_t_given_s, torch.sqrt(sigma2_t_given_s), alpha_t_given_s, alpha_s


def step_coefficients(gamma: torch.Tensor, timesteps: int) -> Dict[str, torch.Tensor]:
    """Per-step scalars of sample_p_zs_given_zt (pharmacodiff.py:387-400,413-420) for every
    s in 0..T-1 (index = s).  All fp32, same op order as the reference."""
    s_int = torch.arange(timesteps)
    s = s_int.float() / timesteps                   # pharmacodiff.py:469
    t = (s_int + 1).float() / timesteps             # pharmacodiff.py:470
    g_s = gamma_lookup(gamma, s, timesteps)
    g_t = gamma_lookup(gamma, t, timesteps)
    s2_ts, s_ts, a_ts, a_s = sigma_and_alpha_t_given_s(g_t, g_s)
    sig_s, sig_t = sigma(g_s), sigma(g_t)
    return {
        "t": t, "s": s,
        "alpha_t_given_s": a_ts,
        "var_terms": s2_ts / a_ts / sig_t,          # pharmacodiff.py:397
        "sigma": s_ts * sig_s / sig_t,              # pharmacodiff.py:400
        # endpoint-parameterisation terms (pharmacodiff.py:414,418)
        "ep_zt": a_ts * (sig_s ** 2) / (sig_t ** 2),
        "ep_pred": a_s * s2_ts / (sig_t ** 2),
    }


def segment_mean(x, ptr):
    """dgl.readout_nodes(op='mean') (pharmacodiff.py:104,442): per-graph mean, 0 for empty."""
    B = ptr.numel() - 1
    out = torch.zeros(B, x.shape[1], dtype=x.dtype)
    for g in range(B):
        a, b = int(ptr[g]), int(ptr[g + 1])
        if b > a:
            out[g] = x[a:b].mean(dim=0)
    return out


def sample_step(sd, cfg, batch: PocketBatch, coef, s_idx: int, prot_x, x_t, h_t, noise_x, noise_h,
                endpoint_param_coord=False, endpoint_param_feat=False, prefix="dynamics."):
    """sample_p_zs_given_zt, pharmacodiff.py:380-431.  Returns (prot_x, x_s, h_s)."""
    B = batch.batch_size
    bidx = batch.batch_idxs()
    t = coef["t"][s_idx].expand(B).contiguous()
    pred_h, pred_x = dynamics_forward(sd, cfg, batch, prot_x, x_t, h_t, t, prefix)
    a_ts = coef["alpha_t_given_s"][s_idx]
    if endpoint_param_coord:
        mu_pos = coef["ep_zt"][s_idx] * x_t + coef["ep_pred"][s_idx] * pred_x
    else:
        mu_pos = x_t / a_ts - coef["var_terms"][s_idx] * pred_x
    if endpoint_param_feat:
        mu_feat = coef["ep_zt"][s_idx] * h_t + coef["ep_pred"][s_idx] * pred_h
    else:
        mu_feat = h_t / a_ts - coef["var_terms"][s_idx] * pred_h
    x_s = mu_pos + coef["sigma"][s_idx] * noise_x
    h_s = mu_feat + coef["sigma"][s_idx] * noise_h
    com = segment_mean(x_s, batch.pharm_ptr)          # com_removal('pharmacophore'), :88-108
    x_s = x_s - com[bidx["pharm"]]
    prot_x = prot_x - com[bidx["prot"]]
    return prot_x, x_s, h_s


def sample_given_receptor(sd, cfg, batch: PocketBatch, n_timesteps: int, precision: float,
                          noise: torch.Tensor, init_pharm_com: Optional[torch.Tensor] = None,
                          pharm_feat_norm_constant: float = 1.0, return_traj: bool = False,
                          endpoint_param_coord=False, endpoint_param_feat=False,
                          n_steps: Optional[int] = None):
    """sample_given_receptor, pharmacodiff.py:433-514, with injected noise.

    noise: [T+1, Nf, 3+pharm_nf]; noise[0] is the initial draw (x columns are drawn before h,
    pharmacodiff.py:455-456), noise[1+i] is the i-th loop iteration's draw (s = T-1-i; x before
    h, pharmacodiff.py:423-424).  ``n_steps`` (test/bench only) stops after that many
    iterations.  Returns (x_0 [Nf,3], h_0 [Nf,pharm_nf]) in the caller's frame (+ frames)."""
    bidx = batch.batch_idxs()
    gamma = gamma_table(n_timesteps, precision)
    coef = step_coefficients(gamma, n_timesteps)
    init_prot_com = segment_mean(batch.prot_x, batch.prot_ptr)       # :442
    if init_pharm_com is None:
        init_pharm_com = init_prot_com                                # :448-449
    prot_x = batch.prot_x - init_pharm_com[bidx["prot"]]             # :452
    x_t = noise[0][:, :3].clone()
    h_t = noise[0][:, 3:].clone()
    frames = []

    def frame(px, xt, ht):
        # get_pos_feat_for_visual, pharmacodiff.py:360-378
        prot_com = segment_mean(px, batch.prot_ptr)
        delta = (init_prot_com - prot_com)[bidx["pharm"]]
        return (xt + delta).clone(), (ht * pharm_feat_norm_constant).clone()

    if return_traj:
        frames.append(frame(prot_x, x_t, h_t))
    it = 0
    for s in reversed(range(n_timesteps)):
        if n_steps is not None and it >= n_steps:
            break
        nz = noise[1 + it]
        prot_x, x_t, h_t = sample_step(sd, cfg, batch, coef, s, prot_x, x_t, h_t, nz[:, :3], nz[:, 3:],
                                       endpoint_param_coord, endpoint_param_feat)
        if return_traj:
            frames.append(frame(prot_x, x_t, h_t))
        it += 1
    # :480-488  rename t->0, remove protein COM, add back initial protein COM, unnormalize
    prot_com = segment_mean(prot_x, batch.prot_ptr)
    x_0 = x_t - prot_com[bidx["pharm"]]
    x_0 = x_0 + init_prot_com[bidx["pharm"]]
    h_0 = h_t * pharm_feat_norm_constant
    if return_traj:
        return x_0, h_0, frames
    return x_0, h_0


def copy_pocket(pocket: PocketBatch, n_centers: int) -> PocketBatch:
    """copy_graph with pharm_feats_per_copy (utils/unorganized_utils.py:28-81) for one copy: same protein nodes and
    pp edges, ``n_centers`` pharmacophore nodes (their features are zeros and are overwritten by the sampler)."""
    assert pocket.batch_size == 1
    return PocketBatch(pocket.prot_x.clone(), pocket.prot_h.clone(), pocket.prot_ptr.clone(),
                       torch.tensor([0, int(n_centers)], dtype=torch.int64), pocket.pp_src.clone(), pocket.pp_dst.clone())


def concat_pockets(pockets: List[PocketBatch]) -> PocketBatch:
    """dgl.batch (pharmacodiff.py:554): node ids and edges concatenated in list order."""
    po, fo, srcs, dsts, pp, fp = 0, 0, [], [], [0], [0]
    for b in pockets:
        srcs.append(b.pp_src + po)
        dsts.append(b.pp_dst + po)
        for g in range(b.batch_size):
            pp.append(po + int(b.prot_ptr[g + 1]))
            fp.append(fo + int(b.pharm_ptr[g + 1]))
        po += int(b.prot_ptr[-1])
        fo += int(b.pharm_ptr[-1])
    return PocketBatch(torch.cat([b.prot_x for b in pockets]), torch.cat([b.prot_h for b in pockets]),
                       torch.tensor(pp, dtype=torch.int64), torch.tensor(fp, dtype=torch.int64),
                       torch.cat(srcs), torch.cat(dsts))


def sample(sd, cfg, ref_pockets: List[PocketBatch], n_pharms: List[List[int]], max_batch_size: int, n_timesteps: int,
           precision: float, noises: List[torch.Tensor], init_pharm_com: Optional[torch.Tensor] = None):
    """PharmacophoreDiff.sample, pharmacodiff.py:516-578: one copy of the pocket per requested pharmacophore
    (:541-545), flattened in pocket order, sampled in batches of ``max_batch_size`` (:551-568) with the pocket's
    row of ``init_pharm_com`` (:556; default: the receptor COMs, :531-535), regrouped per pocket (:570-576).
    ``noises[i]`` holds the draws of batch i.  Returns [[(x_0, h_0) per pharmacophore] per pocket]."""
    if init_pharm_com is None:
        init_pharm_com = torch.stack([p.prot_x.mean(dim=0) for p in ref_pockets], dim=0)
    graphs, ref_idx = [], []
    for r, (pocket, sizes) in enumerate(zip(ref_pockets, n_pharms)):
        graphs.extend(copy_pocket(pocket, n) for n in sizes)
        ref_idx.extend([r] * len(sizes))
    flat = []
    for bi, start in enumerate(range(0, len(graphs), max_batch_size)):
        chunk = graphs[start:start + max_batch_size]
        batch = concat_pockets(chunk)
        coms = init_pharm_com[ref_idx[start:start + max_batch_size]]
        x0, h0 = sample_given_receptor(sd, cfg, batch, n_timesteps, precision, noises[bi], init_pharm_com=coms)
        for g in range(batch.batch_size):
            a, b = int(batch.pharm_ptr[g]), int(batch.pharm_ptr[g + 1])
            flat.append((x0[a:b], h0[a:b]))
    out, end = [], 0
    for sizes in n_pharms:
        start, end = end, end + len(sizes)
        out.append(flat[start:end])
    return out


def training_forward(sd, cfg, batch: PocketBatch, pharm_x0, pharm_h0, n_timesteps: int,
                     precision: float, t_int: torch.Tensor, eps_h: torch.Tensor, eps_x: torch.Tensor,
                     phase: str = "train", pharm_feat_norm_constant: float = 1.0,
                     weighted_loss: bool = False, remove_com: bool = True, dropout=None):
    """PharmacophoreDiff.forward, pharmacodiff.py:162-243 (epsilon parameterisation), with the
    random draws (t_int: :185, eps_h then eps_x: :189-192) injected."""
    bidx = batch.batch_idxs()
    h0 = pharm_h0 / pharm_feat_norm_constant                                   # :168
    com = segment_mean(pharm_x0, batch.pharm_ptr)                               # :179
    x0 = pharm_x0 - com[bidx["pharm"]]
    prot_x = batch.prot_x - com[bidx["prot"]]
    t = t_int.float() / n_timesteps                                             # :185-186
    gamma = gamma_table(n_timesteps, precision)
    gamma_t = gamma_lookup(gamma, t, n_timesteps)
    alpha_t = alpha(gamma_t)[bidx["pharm"]][:, None]
    sigma_t = sigma(gamma_t)[bidx["pharm"]][:, None]
    x_t = alpha_t * x0 + sigma_t * eps_x                                        # :117
    h_t = alpha_t * h0 + sigma_t * eps_h                                        # :118
    if remove_com:                                                              # :124-125
        c = segment_mean(x_t, batch.pharm_ptr)
        x_t = x_t - c[bidx["pharm"]]
        prot_x = prot_x - c[bidx["prot"]]
    h_dyn, x_dyn = dynamics_forward(sd, cfg, batch, prot_x, x_t, h_t, t, dropout=dropout)   # :199
    h_loss = (eps_h - h_dyn).square().sum(dim=1)                                # :208
    h_0_pred = (h_t - sigma_t * h_dyn) / alpha_t                                # :209
    x_loss = (eps_x - x_dyn).square().sum(dim=1)                                # :217
    x_0_pred = (x_t - sigma_t * x_dyn) / alpha_t                                # :219
    weight_metric = 1 - t[bidx["pharm"]]                                        # :221
    weight_loss = weight_metric if weighted_loss else torch.ones_like(weight_metric)
    losses = {
        phase + " pos loss": (x_loss * weight_loss).sum() / eps_x.numel(),      # :231
        phase + " feat loss": (h_loss * weight_loss).sum() / eps_h.numel(),     # :232
    }
    # NOTE :235 compares against g.nodes['pharm'].data['x_0'], which was COM-shifted at :179
    # but not by the second shift at :125.
    err = (x_0_pred - x0).square().sum(dim=1)
    pred_types = h_0_pred.argmax(dim=1)
    hit = (pred_types == h0.argmax(dim=1)).float()
    metrics = {
        phase + " position error": err.mean(),
        phase + " weighted position error": (weight_metric * err).mean(),
        phase + " accuracy": hit.mean(),
        phase + " weighted accuracy": (weight_metric * hit).mean(),
    }
    return losses, metrics


def training_grads(sd, cfg, batch: PocketBatch, pharm_x0, pharm_h0, n_timesteps: int, precision: float,
                   t_int, eps_h, eps_x, dropout=None, weighted_loss: bool = False):
    """One training_step's loss and gradients (pharmacodiff.py:265-276: total loss = pos loss +
    feat loss; backward through PharmacophoreDiff.forward), by torch autograd over the
    restatement above.  Returns (losses, metrics, {key: dLoss/dparam})."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    with torch.enable_grad():
        losses, metrics = training_forward(leaf, cfg, batch, pharm_x0, pharm_h0, n_timesteps, precision,
                                           t_int, eps_h, eps_x, weighted_loss=weighted_loss, dropout=dropout)
        total = losses["train pos loss"] + losses["train feat loss"]
        total.backward()
    grads = {k: (torch.zeros_like(v) if v.grad is None else v.grad.detach()) for k, v in leaf.items()}
    return ({k: v.detach() for k, v in losses.items()}, {k: v.detach() for k, v in metrics.items()}, grads)


def dropout_masks(cfg: DynamicsConfig, n_pharm: int, n_prot: int, p: float, seed: int):
    """Seeded {0, 1/(1-p)} masks in the layout conv_layer takes (one dict per layer)."""
    g = torch.Generator().manual_seed(seed)
    S, V = cfg.n_hidden_scalars, cfg.vector_size
    out = []
    for _ in range(cfg.n_convs):
        d = {}
        for nt, n in (("pharm", n_pharm), ("prot", n_prot)):
            d[nt] = tuple((torch.rand(n, w, generator=g) >= p).float() / (1.0 - p) for w in (S, V, S, V))
        out.append(d)
    return out


# --------------------------------------------------------------------------------------
# synthetic inputs -- shared by tests and bench; deterministic
# --------------------------------------------------------------------------------------
def synthetic_pocket(seed: int, n_prot: int, rec_nf: int = 11):
    """Np points uniform in a ball of density 0.05 atoms/A^3 with a 1.2 A exclusion radius;
    element one-hots with P(C,N,O,S) = (0.62, 0.17, 0.19, 0.02)."""
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


def synthetic_batch(seeds, n_prot, n_pharm, cfg: DynamicsConfig) -> PocketBatch:
    """B graphs; ``n_prot`` / ``n_pharm`` are ints or per-graph lists (ragged)."""
    seeds = list(seeds)
    if isinstance(n_pharm, int):
        n_pharm = [n_pharm] * len(seeds)
    if isinstance(n_prot, int):
        n_prot = [n_prot] * len(seeds)
    xs, hs = [], []
    for sd, npg in zip(seeds, n_prot):
        x, h = synthetic_pocket(sd, npg, cfg.rec_nf)
        xs.append(x)
        hs.append(h)
    prot_ptr = torch.tensor([0] + list(np.cumsum([x.shape[0] for x in xs])), dtype=torch.int64)
    pharm_ptr = torch.tensor([0] + list(np.cumsum(n_pharm)), dtype=torch.int64)
    prot_x = torch.cat(xs)
    src, dst = build_pp_edges(prot_x, prot_ptr, cfg.cutoff_pp, 100)
    return PocketBatch(prot_x, torch.cat(hs), prot_ptr, pharm_ptr, src, dst)


def state_dict_spec(cfg: DynamicsConfig):
    """(name, shape, fan_in) for every tensor of PharmacophoreDiff.state_dict() under
    'dynamics.' in the reference key layout (probed: 244 tensors at dev.yml).
    fan_in == 0 marks LayerNorm weight (ones) / bias (zeros) / the empty dummy parameter."""
    S, V, R = cfg.n_hidden_scalars, cfg.vector_size, cfg.rbf_dim
    spec = []

    def gvp(p, vi, vo, si, so, h=None):
        h = max(vi, vo) if h is None else h
        spec.append((p + "Wh", (vi, h), vi))
        spec.append((p + "Wu", (h, vo), h))
        spec.append((p + "to_feats_out.0.weight", (so, h + si), h + si))
        spec.append((p + "to_feats_out.0.bias", (so,), h + si))
        spec.append((p + "scalar_to_vector_gates.weight", (vo, so), so))
        spec.append((p + "scalar_to_vector_gates.bias", (vo,), so))

    for nt, nf in (("pharm", cfg.pharm_nf), ("prot", cfg.rec_nf)):
        p = f"dynamics.{nt}_encoder."
        spec.append((p + "0.weight", (S, nf + 1), nf + 1))
        spec.append((p + "0.bias", (S,), nf + 1))
        spec.append((p + "2.weight", (S,), 0))
        spec.append((p + "2.bias", (S,), -1))
    for i in range(cfg.n_convs):
        p = f"dynamics.noise_predictor.conv_layers.{i}."
        for et in ETYPES:
            key = "_".join(CANONICAL[et])
            for j in range(cfg.n_message_gvps):
                if j == 0:
                    gvp(f"{p}edge_message_fns.{key}.{j}.", V + 1, V, S + R, S)
                else:
                    gvp(f"{p}edge_message_fns.{key}.{j}.", V, V, S, S)
        for nt in ("pharm", "prot"):
            for j in range(cfg.n_update_gvps):
                gvp(f"{p}node_update_fns.{nt}.{j}.", V, V, S, S)
        for which in ("update_layer_norms", "message_layer_norms"):
            for nt in ("pharm", "prot"):
                spec.append((f"{p}{which}.{nt}.feat_norm.weight", (S,), 0))
                spec.append((f"{p}{which}.{nt}.feat_norm.bias", (S,), -1))
        spec.append((f"{p}dropout.vector_dropout.dummy_param", (0,), -2))
    p = "dynamics.noise_predictor.noise_predictor."
    for k in range(cfg.n_noise_gvps):
        if k == cfg.n_noise_gvps - 1:
            gvp(f"{p}gvps.{k}.", V, 1, S, 64)
        else:
            gvp(f"{p}gvps.{k}.", V, V, S, S)
    spec.append((p + "to_scalar_output.weight", (cfg.pharm_nf, 64), 64))
    spec.append((p + "to_scalar_output.bias", (cfg.pharm_nf,), 64))
    return spec


def make_state_dict(cfg: DynamicsConfig, seed: int = 0, perturb_norm: bool = True) -> Dict[str, torch.Tensor]:
    """Deterministic (numpy PCG64) weights in the reference key layout.  Distributions mimic
    the reference initialisers (uniform +-1/sqrt(fan_in)); LayerNorm affine parameters are
    perturbed away from (1, 0) so that parity tests exercise them.  The same function feeds
    the reference model (via load_state_dict) when goldens are generated, so no weight file
    needs to be committed."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape, fan in state_dict_spec(cfg):
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
