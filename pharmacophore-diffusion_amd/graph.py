"""DGL-free batch container for pocket + pharmacophore graphs and the batch bookkeeping helpers
of the reference (pharmacoforge/utils/unorganized_utils.py:6-95,
pharmacoforge/dataset/protein_pharm_dataset.py:210-271).

A ``PocketGraph`` is one graph or a batch of graphs stored as flat tensors: node ids are local to
their node type ('prot', 'pharm', 'prot_ph'); graph g owns the contiguous ranges given by the
``*_ptr`` vectors.  Only the static prot->prot ('pp') edges are stored: the pharm<->pharm and
prot<->pharm edges are rebuilt on the device at every denoising step (dynamics_gvp.py:187-246)."""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional

import torch


_pocket_uids = itertools.count(1)


def _ptr(counts) -> torch.Tensor:
    c = torch.as_tensor(counts, dtype=torch.int64).reshape(-1)
    return torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(c, 0)])


def _ptr1(n: int) -> torch.Tensor:             # _ptr([n]) in one tensor construction (unbatch: three per graph)
    return torch.tensor([0, int(n)], dtype=torch.int64)


@dataclass
class PocketGraph:
    prot_x: torch.Tensor                       # [Np,3]  'prot'.x_0
    prot_h: torch.Tensor                       # [Np,rec_nf] 'prot'.h_0
    prot_ptr: torch.Tensor                     # [B+1] int64 (host)
    pharm_ptr: torch.Tensor                    # [B+1] int64 (host)
    pp_src: torch.Tensor                       # [Epp] int64 (host), prot-local over the batch
    pp_dst: torch.Tensor
    pharm_x0: Optional[torch.Tensor] = None    # [Nf,3]   'pharm'.x_0 (reference pharmacophore / zeros)
    pharm_h0: Optional[torch.Tensor] = None    # [Nf,pharm_nf]
    prot_ph_x: Optional[torch.Tensor] = None   # receptor pharmacophore features ('prot_ph' nodes)
    prot_ph_h: Optional[torch.Tensor] = None
    prot_ph_ptr: Optional[torch.Tensor] = None
    x_t: Optional[torch.Tensor] = None         # 'pharm'.x_t / h_t during sampling
    h_t: Optional[torch.Tensor] = None
    pp_ptr: Optional[torch.Tensor] = None      # [B+1] int64 (host): graph g owns pp edges [pp_ptr[g], pp_ptr[g+1]) -- set by
                                               # batch(); lets unbatch() skip the search for the per-graph edge ranges
    pocket_uid: Optional[torch.Tensor] = None  # [B] int64 (host): graphs with the same value are COPIES of one pocket (same
                                               # atoms, features, coordinates, pp edges) -- stamped by copy_graph, carried
                                               # through batch / unbatch / to; the engine lets such copies share work

    # -- DGL-like accessors used by the reference's drivers --------------------------------
    @property
    def batch_size(self) -> int:
        return int(self.prot_ptr.numel() - 1)

    @property
    def device(self) -> torch.device:
        return self.prot_x.device

    def num_nodes(self, ntype: str) -> int:
        return int(self._ptr_of(ntype)[-1])

    def batch_num_nodes(self, ntype: str) -> torch.Tensor:
        p = self._ptr_of(ntype)
        return p[1:] - p[:-1]

    def _ptr_of(self, ntype: str) -> torch.Tensor:
        if ntype == "prot":
            return self.prot_ptr
        if ntype == "pharm":
            return self.pharm_ptr
        if ntype == "prot_ph":
            return self.prot_ph_ptr if self.prot_ph_ptr is not None else torch.zeros(self.batch_size + 1, dtype=torch.int64)
        raise KeyError(ntype)

    def to(self, device) -> "PocketGraph":
        def mv(t):
            return None if t is None else t.to(device)
        out = replace(self, prot_x=mv(self.prot_x), prot_h=mv(self.prot_h), pharm_x0=mv(self.pharm_x0),
                      pharm_h0=mv(self.pharm_h0), prot_ph_x=mv(self.prot_ph_x), prot_ph_h=mv(self.prot_ph_h),
                      x_t=mv(self.x_t), h_t=mv(self.h_t))
        if "_i32_cache" in self.__dict__:          # the index tensors are shared with the copy
            out.__dict__["_i32_cache"] = self.__dict__["_i32_cache"]
        return out

    def index_arrays_i32(self):
        """(prot_ptr, pharm_ptr, pp_src, pp_dst) as contiguous int32 numpy arrays -- the form pf_set_pocket_batch takes.
        Made once per graph object (batch() does it at collate time, i.e. in the data loader and off the training step's
        critical path: the int64 -> int32 conversion of ~1.3 M edge indices costs 0.3 ms per bind otherwise) and dropped
        when an index tensor is replaced or written in place."""
        idx = (self.prot_ptr, self.pharm_ptr, self.pp_src, self.pp_dst)
        vers = tuple(t._version for t in idx)
        c = self.__dict__.get("_i32_cache")
        # the entry holds the tensors themselves and is matched by identity: a replaced index tensor that happens to be
        # allocated at a recycled address with the same length cannot hit a stale entry
        if c is None or c[1] != vers or any(a is not b for a, b in zip(c[0], idx)):
            import numpy as np
            c = (idx, vers, tuple(np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.int32) for t in idx))
            self.__dict__["_i32_cache"] = c
        return c[2]

    def batch_idxs(self) -> Dict[str, torch.Tensor]:
        return get_batch_idxs(self)


def get_batch_idxs(g: PocketGraph) -> Dict[str, torch.Tensor]:
    """utils/unorganized_utils.py:83-95: graph id of every node, per node type (on g.device).  Cached on the graph
    object (the ptr arrays of a PocketGraph are immutable): a training loop asks for it every step."""
    key = (str(g.device), g.prot_ptr.data_ptr(), g.pharm_ptr.data_ptr(), int(g.prot_ptr[-1]), int(g.pharm_ptr[-1]))
    cached = g.__dict__.get("_bidx_cache")
    if cached is not None and cached[0] == key:
        return dict(cached[1])
    ar = torch.arange(g.batch_size)
    out = {}
    for nt in ("prot", "pharm", "prot_ph"):
        out[nt] = ar.repeat_interleave(g.batch_num_nodes(nt)).to(g.device)
    g.__dict__["_bidx_cache"] = (key, out)
    return dict(out)


def get_batch_info(g: PocketGraph):
    """utils/unorganized_utils.py:6-15 (node counts per graph; pp edge counts per graph)."""
    nodes = {nt: g.batch_num_nodes(nt) for nt in ("prot", "pharm", "prot_ph")}
    gid = torch.searchsorted(g.prot_ptr[1:].contiguous(), g.pp_dst, right=True)
    edges = {("prot", "pp", "prot"): torch.bincount(gid, minlength=g.batch_size)}
    return nodes, edges


def batch(graphs: List[PocketGraph]) -> PocketGraph:
    """dgl.batch for PocketGraphs (collate_fn, dataset/protein_pharm_dataset.py:268-271)."""
    def cat(name):
        vals = [getattr(g, name) for g in graphs]
        if any(v is None for v in vals):
            return None
        return torch.cat(vals)
    prot_counts = torch.cat([g.batch_num_nodes("prot") for g in graphs])
    pharm_counts = torch.cat([g.batch_num_nodes("pharm") for g in graphs])
    ph_counts = torch.cat([g.batch_num_nodes("prot_ph") for g in graphs])
    srcs, dsts, off, e_counts = [], [], 0, []
    for g in graphs:
        srcs.append(g.pp_src + off)
        dsts.append(g.pp_dst + off)
        off += g.num_nodes("prot")
        e_counts.append(int(g.pp_src.numel()))
    # per-graph edge ranges are known only when every input is a single graph (a batched input keeps its own grouping)
    pp_ptr = _ptr(e_counts) if all(g.batch_size == 1 for g in graphs) else None
    # graphs without a stamp are pockets of their own: fresh values (negative, so they never meet a copy_graph stamp)
    uid = None
    if any(g.pocket_uid is not None for g in graphs):
        uid = torch.cat([g.pocket_uid if g.pocket_uid is not None else -((next(_pocket_uids) << 20) + torch.arange(g.batch_size))
                         for g in graphs])
    out = PocketGraph(cat("prot_x"), cat("prot_h"), _ptr(prot_counts), _ptr(pharm_counts), torch.cat(srcs), torch.cat(dsts),
                      cat("pharm_x0"), cat("pharm_h0"), cat("prot_ph_x"), cat("prot_ph_h"), _ptr(ph_counts),
                      cat("x_t"), cat("h_t"), pp_ptr, uid)
    out.index_arrays_i32()          # the engine's int32 view of the index arrays, made here (collate time)
    return out


def unbatch(g: PocketGraph) -> List[PocketGraph]:
    """dgl.unbatch: per-graph views.  The pp edges of a batched graph are grouped by graph (batch() concatenates them
    in graph order), so each graph's edges are one slice; an arbitrary edge order falls back to masks."""
    out = []
    B = g.batch_size
    if g.pp_ptr is not None and g.pp_ptr.numel() == B + 1 and int(g.pp_ptr[-1]) == g.pp_src.numel():
        grouped, gid, e_ptr = True, None, g.pp_ptr.tolist()
    else:
        gid = torch.searchsorted(g.prot_ptr[1:].contiguous(), g.pp_dst, right=True)
        grouped = bool((gid[1:] >= gid[:-1]).all()) if gid.numel() > 1 else True
        if grouped:
            e_ptr = torch.zeros(B + 1, dtype=torch.int64)
            e_ptr[1:] = torch.cumsum(torch.bincount(gid, minlength=B)[:B], 0)
            e_ptr = e_ptr.tolist()
    pp, fp = g.prot_ptr.tolist(), g.pharm_ptr.tolist()
    have_ph = g.prot_ph_ptr is not None and g.prot_ph_x is not None
    qp = g.prot_ph_ptr.tolist() if have_ph else None

    def sl(t, a, e):
        return None if t is None else t[a:e]
    for b in range(B):
        p0, p1, f0, f1 = pp[b], pp[b + 1], fp[b], fp[b + 1]
        if grouped:
            src, dst = g.pp_src[e_ptr[b]:e_ptr[b + 1]] - p0, g.pp_dst[e_ptr[b]:e_ptr[b + 1]] - p0
        else:
            m = gid == b
            src, dst = g.pp_src[m] - p0, g.pp_dst[m] - p0
        q0, q1 = (qp[b], qp[b + 1]) if have_ph else (0, 0)
        out.append(PocketGraph(g.prot_x[p0:p1], g.prot_h[p0:p1], _ptr1(p1 - p0), _ptr1(f1 - f0), src, dst,
                               sl(g.pharm_x0, f0, f1), sl(g.pharm_h0, f0, f1),
                               sl(g.prot_ph_x, q0, q1), sl(g.prot_ph_h, q0, q1), _ptr1(q1 - q0),
                               sl(g.x_t, f0, f1), sl(g.h_t, f0, f1),
                               pocket_uid=None if g.pocket_uid is None else g.pocket_uid[b:b + 1]))
    return out


def copy_graph(g: PocketGraph, n_copies: int, pharm_feats_per_copy=None, batched_graph: bool = False) -> List[PocketGraph]:
    """utils/unorganized_utils.py:28-81: n copies of a graph; when ``pharm_feats_per_copy`` is given
    copy i gets that many pharmacophore nodes with zeroed features (always indexed from 0, like the
    reference, SURVEY.md section 7 "quirks")."""
    copies = []
    # the copies of a single pocket are stamped as such (a batched input keeps whatever stamps its graphs carry)
    uid = g.pocket_uid if (g.pocket_uid is not None or g.batch_size != 1) else torch.tensor([next(_pocket_uids)], dtype=torch.int64)
    for i in range(n_copies):
        c = replace(g, prot_x=g.prot_x.detach().clone(), prot_h=g.prot_h.detach().clone(), pocket_uid=uid)
        if pharm_feats_per_copy is not None:
            if batched_graph:
                raise ValueError("pharm_feats_per_copy needs a single (unbatched) graph")
            n = int(pharm_feats_per_copy[i])
            nf = g.pharm_h0.shape[1] if g.pharm_h0 is not None else 0
            c.pharm_ptr = _ptr([n])
            c.pharm_x0 = torch.zeros(n, 3, device=g.device)
            c.pharm_h0 = torch.zeros(n, nf, device=g.device)
            c.x_t = c.h_t = None
        else:
            for name in ("pharm_x0", "pharm_h0", "x_t", "h_t"):
                v = getattr(g, name)
                setattr(c, name, None if v is None else v.detach().clone())
        copies.append(c)
    return copies


_edge_engines: dict = {}


def radius_graph_pp(prot_x: torch.Tensor, prot_ptr: torch.Tensor, cutoff: float, max_num_neighbors: int = 100):
    """Static prot->prot radius graph on the GPU (pf_build_pp_edges).  Returns (src, dst) int64 on
    the host, target-major.  Needs the HIP library and a device (no CPU fallback)."""
    from .engine import PfEngine
    key = float(cutoff)
    if key not in _edge_engines:
        _edge_engines[key] = PfEngine(graph_cutoffs={"pp": key})
    return _edge_engines[key].build_pp_edges(prot_x, prot_ptr, max_num_neighbors)


def build_initial_complex_graph(prot_atom_positions: torch.Tensor, prot_atom_features: torch.Tensor, cutoffs: dict,
                                pharm_atom_positions: torch.Tensor = None, pharm_atom_features: torch.Tensor = None,
                                prot_ph_pos: torch.Tensor = None, prot_ph_feat: torch.Tensor = None,
                                pp_edges=None) -> PocketGraph:
    """dataset/protein_pharm_dataset.py:210-266 without DGL.  ``pp_edges`` (src, dst) may be given
    to skip the radius search (e.g. edges stored with a processed dataset)."""
    if (pharm_atom_positions is not None) ^ (pharm_atom_features is not None):
        raise ValueError('pharmacophore position and features must be either be both supplied or both left as None')
    n_prot = prot_atom_positions.shape[0]
    n_pharm = 0 if pharm_atom_positions is None else pharm_atom_positions.shape[0]
    if pp_edges is not None:
        src, dst = pp_edges
    elif cutoffs['pp'] > 0:
        src, dst = radius_graph_pp(prot_atom_positions, _ptr([n_prot]), cutoffs['pp'], 100)
    else:
        src = dst = torch.zeros(0, dtype=torch.int64)
    if prot_ph_pos is not None:
        assert prot_ph_feat is not None
    n_ph = 0 if prot_ph_pos is None else prot_ph_pos.shape[0]
    return PocketGraph(prot_atom_positions, prot_atom_features, _ptr([n_prot]), _ptr([n_pharm]),
                       src.cpu().long(), dst.cpu().long(), pharm_atom_positions, pharm_atom_features,
                       prot_ph_pos, prot_ph_feat, _ptr([n_ph]))


def from_dgl(g) -> PocketGraph:
    """Adapter for callers that still hold a DGL heterograph built by the reference's dataset code
    (only usable where ``dgl`` is importable)."""
    def data(nt, k):
        d = g.nodes[nt].data
        return d[k] if k in d else None
    u, v = g.edges(form='uv', etype='pp')
    have_ph = 'prot_ph' in g.ntypes
    return PocketGraph(data('prot', 'x_0'), data('prot', 'h_0'), _ptr(g.batch_num_nodes('prot').cpu()),
                       _ptr(g.batch_num_nodes('pharm').cpu()), u.cpu().long(), v.cpu().long(),
                       data('pharm', 'x_0'), data('pharm', 'h_0'),
                       data('prot_ph', 'x_0') if have_ph else None, data('prot_ph', 'h_0') if have_ph else None,
                       _ptr(g.batch_num_nodes('prot_ph').cpu()) if have_ph else None,
                       data('pharm', 'x_t'), data('pharm', 'h_t'))


def as_pocket_graph(g) -> PocketGraph:
    return g if isinstance(g, PocketGraph) else from_dgl(g)
