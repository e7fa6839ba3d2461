"""Functional stand-ins for the third-party Python packages the reference imports but this
image lacks (dgl, dgl.function, torch_cluster, torch_scatter, pytorch_lightning, rdkit).

TEST TOOLING, authored by this repo (no reference code).  Purpose: let the reference's own
``pharmacoforge.models.*`` modules be imported from /root/reference *in the build container*
so that ``make_golden.py`` can record their outputs as golden vectors.  Only the ~25 entry
points the hot path touches are provided, with the libraries' documented semantics
(SURVEY.md section 8(c)):

  dgl: heterograph container with per-type node/edge frames, local_scope (feature writes are
       reverted, structure is not), apply_edges (builtin u_sub_v or UDF), multi_update_all
       (per-etype copy_e + sum|mean reducer, zero for in-degree-0 nodes, cross_reducer 'sum'),
       add_edges / remove_edges / edges(form=...), batch / unbatch, readout_nodes(op='mean'),
       batch_num_nodes / batch_num_edges / set_batch_num_*.
  torch_cluster: radius, radius_graph, knn, knn_graph (brute force, python loops; written
       independently of oracle/pf_oracle.py so the two cross-check each other).

Nothing here travels to the GPU box as part of a test: the goldens it produces do.
"""
from __future__ import annotations

import contextlib
import sys
import types
from typing import Dict, List, Tuple

import torch
import torch.nn as nn


# ---------------------------------------------------------------------------------------
# dgl.function
# ---------------------------------------------------------------------------------------
class _USubV:
    def __init__(self, lhs, rhs, out):
        self.lhs, self.rhs, self.out = lhs, rhs, out


class _CopyE:
    def __init__(self, e, out):
        self.e, self.out = e, out


class _Reduce:
    def __init__(self, kind, msg, out):
        self.kind, self.msg, self.out = kind, msg, out


def _fn_sum(msg, out):
    return _Reduce("sum", msg, out)


def _fn_mean(msg, out):
    return _Reduce("mean", msg, out)


# ---------------------------------------------------------------------------------------
# heterograph
# ---------------------------------------------------------------------------------------
class _Frame:
    def __init__(self, store: dict):
        self.data = store


class _View:
    def __init__(self, g, kind):
        self._g, self._kind = g, kind

    def __getitem__(self, key):
        if self._kind == "n":
            return _Frame(self._g._ndata[key])
        return _Frame(self._g._edata[self._g._canon(key)])


class _EdgeBatch:
    def __init__(self, g, cet):
        s_nt, _, d_nt = cet
        src, dst = g._edges[cet]
        self.canonical_etype = cet
        self.data = g._edata[cet]
        self.src = {k: v[src] for k, v in g._ndata[s_nt].items()}
        self.dst = {k: v[dst] for k, v in g._ndata[d_nt].items()}


class HeteroGraph:
    def __init__(self, data_dict, num_nodes_dict, device="cpu"):
        self.ntypes = sorted(num_nodes_dict.keys())
        self._num_nodes = {k: int(v) for k, v in num_nodes_dict.items()}
        self.canonical_etypes = sorted(data_dict.keys())
        self._edges = {}
        for cet, (u, v) in data_dict.items():
            u = torch.as_tensor(u, dtype=torch.int64).reshape(-1)
            v = torch.as_tensor(v, dtype=torch.int64).reshape(-1)
            self._edges[cet] = (u.clone(), v.clone())
        self._ndata = {nt: {} for nt in self.ntypes}
        self._edata = {cet: {} for cet in self.canonical_etypes}
        self._bnn = {nt: torch.tensor([self._num_nodes[nt]]) for nt in self.ntypes}
        self._bne = {cet: torch.tensor([self._edges[cet][0].numel()]) for cet in self.canonical_etypes}
        self.device = torch.device(device)

    # -- helpers
    def _canon(self, et):
        if isinstance(et, tuple):
            return et
        for cet in self.canonical_etypes:
            if cet[1] == et:
                return cet
        raise KeyError(et)

    @property
    def nodes(self):
        return _View(self, "n")

    @property
    def edges(self):
        return _EdgesAccessor(self)

    @property
    def batch_size(self):
        return int(next(iter(self._bnn.values())).numel())

    def num_nodes(self, ntype=None):
        return self._num_nodes[ntype]

    def num_edges(self, etype):
        return int(self._edges[self._canon(etype)][0].numel())

    def batch_num_nodes(self, ntype):
        return self._bnn[ntype]

    def batch_num_edges(self, etype):
        return self._bne[self._canon(etype)]

    def set_batch_num_nodes(self, d):
        for k, v in d.items():
            self._bnn[k] = v

    def set_batch_num_edges(self, d):
        for k, v in d.items():
            self._bne[self._canon(k)] = v

    def to(self, device):
        return self

    @contextlib.contextmanager
    def local_scope(self):
        nsnap = {k: dict(v) for k, v in self._ndata.items()}
        esnap = {k: dict(v) for k, v in self._edata.items()}
        try:
            yield
        finally:
            self._ndata = nsnap
            self._edata = esnap

    # -- structure mutation
    def add_edges(self, u, v, etype=None):
        cet = self._canon(etype)
        u = torch.as_tensor(u, dtype=torch.int64).reshape(-1)
        v = torch.as_tensor(v, dtype=torch.int64).reshape(-1)
        s, d = self._edges[cet]
        self._edges[cet] = (torch.cat([s, u]), torch.cat([d, v]))
        # documented: existing edge features are extended with zeros for the new edges
        self._edata[cet] = {k: torch.cat([t, torch.zeros((u.numel(),) + tuple(t.shape[1:]), dtype=t.dtype)])
                            for k, t in self._edata[cet].items()}
        # like DGL, a structure change drops batch bookkeeping for that etype to "one graph"
        self._bne[cet] = torch.tensor([self._edges[cet][0].numel()])

    def remove_edges(self, eids, etype=None):
        cet = self._canon(etype)
        s, d = self._edges[cet]
        keep = torch.ones(s.numel(), dtype=torch.bool)
        keep[torch.as_tensor(eids, dtype=torch.int64)] = False
        self._edges[cet] = (s[keep], d[keep])
        self._edata[cet] = {k: v[keep] for k, v in self._edata[cet].items()}
        self._bne[cet] = torch.tensor([self._edges[cet][0].numel()])

    # -- message passing
    def apply_edges(self, func, etype=None):
        cet = self._canon(etype)
        s_nt, _, d_nt = cet
        src, dst = self._edges[cet]
        if isinstance(func, _USubV):
            self._edata[cet][func.out] = self._ndata[s_nt][func.lhs][src] - self._ndata[d_nt][func.rhs][dst]
            return
        out = func(_EdgeBatch(self, cet))
        self._edata[cet].update(out)

    def _reduce_one(self, cet, msg_fn, red):
        _, _, d_nt = cet
        _, dst = self._edges[cet]
        m = self._edata[cet][msg_fn.e]
        n = self._num_nodes[d_nt]
        out = torch.zeros((n,) + tuple(m.shape[1:]), dtype=m.dtype)
        if dst.numel():
            out.index_add_(0, dst, m)
            if red.kind == "mean":
                deg = torch.bincount(dst, minlength=n).clamp(min=1).to(m.dtype)
                out = out / deg.view(-1, *([1] * (m.dim() - 1)))
        return d_nt, red.out, out

    def multi_update_all(self, etype_dict, cross_reducer="sum"):
        assert cross_reducer == "sum"
        acc: Dict[Tuple[str, str], torch.Tensor] = {}
        for et, (msg_fn, red) in etype_dict.items():
            d_nt, name, out = self._reduce_one(self._canon(et), msg_fn, red)
            key = (d_nt, name)
            acc[key] = out if key not in acc else acc[key] + out
        for (d_nt, name), v in acc.items():
            self._ndata[d_nt][name] = v

    def update_all(self, msg_fn, red, etype=None):
        d_nt, name, out = self._reduce_one(self._canon(etype), msg_fn, red)
        self._ndata[d_nt][name] = out


class _EdgesAccessor:
    """g.edges[etype].data  and  g.edges(form=..., etype=...)"""

    def __init__(self, g):
        self._g = g

    def __getitem__(self, key):
        return _Frame(self._g._edata[self._g._canon(key)])

    def __call__(self, form="uv", etype=None):
        s, d = self._g._edges[self._g._canon(etype)]
        if form == "eid":
            return torch.arange(s.numel())
        if form == "uv":
            return s, d
        raise ValueError(form)


def heterograph(data_dict, num_nodes_dict=None, device="cpu"):
    return HeteroGraph(data_dict, num_nodes_dict, device)


def batch(graphs: List[HeteroGraph]) -> HeteroGraph:
    g0 = graphs[0]
    nn_tot = {nt: sum(g._num_nodes[nt] for g in graphs) for nt in g0.ntypes}
    data = {}
    for cet in g0.canonical_etypes:
        s_nt, _, d_nt = cet
        so = do = 0
        us, vs = [], []
        for g in graphs:
            u, v = g._edges[cet]
            us.append(u + so)
            vs.append(v + do)
            so += g._num_nodes[s_nt]
            do += g._num_nodes[d_nt]
        data[cet] = (torch.cat(us), torch.cat(vs))
    out = HeteroGraph(data, nn_tot)
    for nt in g0.ntypes:
        for k in g0._ndata[nt].keys():
            out._ndata[nt][k] = torch.cat([g._ndata[nt][k] for g in graphs])
        out._bnn[nt] = torch.cat([g._bnn[nt] for g in graphs])
    for cet in g0.canonical_etypes:
        for k in g0._edata[cet].keys():
            out._edata[cet][k] = torch.cat([g._edata[cet][k] for g in graphs])
        out._bne[cet] = torch.cat([g._bne[cet] for g in graphs])
    return out


def unbatch(g: HeteroGraph) -> List[HeteroGraph]:
    B = g.batch_size
    outs = []
    noff = {nt: 0 for nt in g.ntypes}
    eoff = {cet: 0 for cet in g.canonical_etypes}
    for b in range(B):
        nn_b = {nt: int(g._bnn[nt][b]) for nt in g.ntypes}
        data = {}
        for cet in g.canonical_etypes:
            s_nt, _, d_nt = cet
            ne = int(g._bne[cet][b])
            u, v = g._edges[cet]
            data[cet] = (u[eoff[cet]:eoff[cet] + ne] - noff[s_nt], v[eoff[cet]:eoff[cet] + ne] - noff[d_nt])
        gi = HeteroGraph(data, nn_b)
        for nt in g.ntypes:
            for k, val in g._ndata[nt].items():
                gi._ndata[nt][k] = val[noff[nt]:noff[nt] + nn_b[nt]]
        for cet in g.canonical_etypes:
            ne = int(g._bne[cet][b])
            for k, val in g._edata[cet].items():
                gi._edata[cet][k] = val[eoff[cet]:eoff[cet] + ne]
            eoff[cet] += ne
        for nt in g.ntypes:
            noff[nt] += nn_b[nt]
        outs.append(gi)
    return outs


def readout_nodes(g: HeteroGraph, feat, ntype=None, op="mean"):
    assert op == "mean"
    x = g._ndata[ntype][feat]
    cnt = g._bnn[ntype]
    out = torch.zeros((cnt.numel(),) + tuple(x.shape[1:]), dtype=x.dtype)
    o = 0
    for b in range(cnt.numel()):
        n = int(cnt[b])
        if n:
            out[b] = x[o:o + n].mean(dim=0)
        o += n
    return out


# ---------------------------------------------------------------------------------------
# torch_cluster (brute force, independent of the oracle's vectorised versions)
# ---------------------------------------------------------------------------------------
def _sqdist(a, b):
    d = a - b
    sq = d * d
    return float((sq[0] + sq[1]) + sq[2])     # fp32 arithmetic on 0-d tensors, then exact widening


def _batch_or_zero(batch, n):
    return torch.zeros(n, dtype=torch.int64) if batch is None else batch


def tc_radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32):
    bx, by = _batch_or_zero(batch_x, x.shape[0]), _batch_or_zero(batch_y, y.shape[0])
    r2 = float(torch.tensor(r, dtype=torch.float32) ** 2)
    rows, cols = [], []
    for i in range(y.shape[0]):
        c = 0
        for j in range(x.shape[0]):
            if int(bx[j]) != int(by[i]):
                continue
            if _sqdist(x[j], y[i]) < r2:
                if c >= max_num_neighbors:
                    break
                rows.append(i)
                cols.append(j)
                c += 1
    return torch.tensor([rows, cols], dtype=torch.int64).reshape(2, -1)


def tc_radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target"):
    b = _batch_or_zero(batch, x.shape[0])
    r2 = float(torch.tensor(r, dtype=torch.float32) ** 2)
    src, dst = [], []
    for i in range(x.shape[0]):
        c = 0
        for j in range(x.shape[0]):
            if i == j or int(b[i]) != int(b[j]):
                continue
            if _sqdist(x[j], x[i]) < r2:
                if c >= max_num_neighbors:
                    break
                src.append(j)
                dst.append(i)
                c += 1
    return torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1)


def tc_knn(x, y, k, batch_x=None, batch_y=None):
    bx, by = _batch_or_zero(batch_x, x.shape[0]), _batch_or_zero(batch_y, y.shape[0])
    rows, cols = [], []
    for i in range(y.shape[0]):
        cand = [(_sqdist(x[j], y[i]), j) for j in range(x.shape[0]) if int(bx[j]) == int(by[i])]
        cand.sort()
        for _, j in cand[:k]:
            rows.append(i)
            cols.append(j)
    return torch.tensor([rows, cols], dtype=torch.int64).reshape(2, -1)


def tc_knn_graph(x, k, batch=None, loop=False, flow="source_to_target"):
    b = _batch_or_zero(batch, x.shape[0])
    src, dst = [], []
    for i in range(x.shape[0]):
        cand = [(_sqdist(x[j], x[i]), j) for j in range(x.shape[0]) if j != i and int(b[j]) == int(b[i])]
        cand.sort()
        for _, j in cand[:k]:
            src.append(j)
            dst.append(i)
    return torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1)


# ---------------------------------------------------------------------------------------
# pytorch_lightning
# ---------------------------------------------------------------------------------------
class LightningModule(nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass

    @property
    def device(self):
        return torch.device("cpu")


def install(reference_root: str = "/root/reference"):
    """Register the stand-ins in sys.modules and put the reference on sys.path."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    fn = mod("dgl.function", u_sub_v=_USubV, copy_e=_CopyE, sum=_fn_sum, mean=_fn_mean)
    class DGLDataset:                 # dgl.data.DGLDataset as far as the reference's dataset class uses it: a name, nothing else
        def __init__(self, name=None, **kwargs):
            self._name = name

    data = mod("dgl.data", DGLDataset=DGLDataset)
    dl = mod("dgl.dataloading", GraphDataLoader=object)
    mod("dgl", DGLHeteroGraph=HeteroGraph, heterograph=heterograph, batch=batch, unbatch=unbatch,
        readout_nodes=readout_nodes, function=fn, data=data, dataloading=dl)
    mod("torch_cluster", radius=tc_radius, radius_graph=tc_radius_graph, knn=tc_knn, knn_graph=tc_knn_graph)
    mod("torch_scatter", segment_coo=None, segment_csr=None)
    mod("pytorch_lightning", LightningModule=LightningModule)
    chem = mod("rdkit.Chem")
    mod("rdkit", Chem=chem)
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
