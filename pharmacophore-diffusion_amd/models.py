"""Host-side mirror of the reference's model classes for the denoising path.

The nn.Module tree below reproduces the reference's *parameter layout* (so ``state_dict()`` /
``load_state_dict()`` / Lightning ``.ckpt`` files use exactly the reference key set, SURVEY.md
section 5) and its *call surface*:

    PharmRecDynamicsGVP.forward(g, timestep, batch_idxs)      pharmacoforge/models/dynamics_gvp.py:131
    PharmacophoreDiff.sample_given_receptor / sample / forward  pharmacoforge/models/pharmacodiff.py:433,516,162

but none of these modules computes anything in PyTorch: every forward -- and, in training, the
backward of the dynamics -- goes through libpfdyn.so (csrc/, C ABI in include/pfdyn.h); PyTorch
supplies the autograd plumbing around it, the elementwise loss and the optimiser.  There is no
eager / CPU fallback -- without the HIP library and a GPU the calls raise.
"""
from __future__ import annotations

import math
from math import ceil
from pathlib import Path
from typing import Dict, List, Optional, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .analysis import SampleAnalyzer, SampledPharmacophore
from .engine import PfEngine
from .graph import PocketGraph, as_pocket_graph, batch as batch_graphs, copy_graph, get_batch_idxs, unbatch
from .schedule import PredefinedNoiseSchedule, alpha as _alpha, sigma as _sigma, sigma_and_alpha_t_given_s, step_coefficients

try:  # subclass LightningModule when Lightning is importable (drop-in for train.py / load_from_checkpoint)
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # Lightning absent (this image): same surface on top of nn.Module
    pl = None
    _Base = nn.Module


# ---------------------------------------------------------------------------------------------
# parameter containers with the reference's names (gvp.py:43-166, 343-437; dynamics_gvp.py:10-129)
# ---------------------------------------------------------------------------------------------
class GVP(nn.Module):
    def __init__(self, dim_vectors_in, dim_vectors_out, dim_feats_in, dim_feats_out, hidden_vectors=None,
                 feats_activation=None, vectors_activation=None, vector_gating=True, xavier_init=False):
        super().__init__()
        if not vector_gating:
            raise NotImplementedError("only vector-gated GVPs are on the hot path (gvp.py:79-80)")
        self.dim_vectors_in, self.dim_feats_in, self.dim_vectors_out = dim_vectors_in, dim_feats_in, dim_vectors_out
        dim_h = max(dim_vectors_in, dim_vectors_out) if hidden_vectors is None else hidden_vectors
        wh_k, wu_k = 1 / math.sqrt(dim_vectors_in), 1 / math.sqrt(dim_h)
        self.Wh = nn.Parameter(torch.zeros(dim_vectors_in, dim_h).uniform_(-wh_k, wh_k))
        self.Wu = nn.Parameter(torch.zeros(dim_h, dim_vectors_out).uniform_(-wu_k, wu_k))
        self.to_feats_out = nn.Sequential(nn.Linear(dim_h + dim_feats_in, dim_feats_out), nn.SiLU())
        self.scalar_to_vector_gates = nn.Linear(dim_feats_out, dim_vectors_out)
        if xavier_init:
            nn.init.xavier_uniform_(self.scalar_to_vector_gates.weight, gain=1)
            nn.init.constant_(self.scalar_to_vector_gates.bias, 0)


class _VDropout(nn.Module):
    def __init__(self, drop_rate):
        super().__init__()
        self.drop_rate = drop_rate
        self.dummy_param = nn.Parameter(torch.empty(0))      # the reference's empty tensor, kept for key parity


class GVPDropout(nn.Module):
    def __init__(self, rate):
        super().__init__()
        self.vector_dropout = _VDropout(rate)
        self.feat_dropout = nn.Dropout(rate)


class GVPLayerNorm(nn.Module):
    def __init__(self, feats_h_size, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.feat_norm = nn.LayerNorm(feats_h_size)


class GVPMultiEdgeConv(nn.Module):
    def __init__(self, etypes, scalar_size=128, vector_size=16, n_message_gvps=1, n_update_gvps=1, rbf_dmax=15,
                 rbf_dim=16, message_norm: Union[float, str, Dict] = 10, dropout=0.0):
        super().__init__()
        self.etypes = etypes
        self.dst_ntypes = sorted(set(e[2] for e in etypes))
        if isinstance(message_norm, str) and message_norm != 'mean':
            raise ValueError(f"message_norm values must be 'mean' or a positive number, got {message_norm}")
        if isinstance(message_norm, (int, float)) and message_norm < 0:
            raise ValueError(f"message_norm values must be 'mean' or a positive number, got {message_norm}")
        self.edge_message_fns = nn.ModuleDict()
        for etype in etypes:
            gvps = []
            for i in range(n_message_gvps):
                vi = vector_size + 1 if i == 0 else vector_size
                si = scalar_size + rbf_dim if i == 0 else scalar_size
                gvps.append(GVP(vi, vector_size, si, scalar_size))
            self.edge_message_fns['_'.join(etype)] = nn.Sequential(*gvps)
        self.node_update_fns = nn.ModuleDict()
        self.update_layer_norms = nn.ModuleDict()
        self.message_layer_norms = nn.ModuleDict()
        for ntype in self.dst_ntypes:
            self.node_update_fns[ntype] = nn.Sequential(*[GVP(vector_size, vector_size, scalar_size, scalar_size)
                                                          for _ in range(n_update_gvps)])
            self.message_layer_norms[ntype] = GVPLayerNorm(scalar_size)
            self.update_layer_norms[ntype] = GVPLayerNorm(scalar_size)
        self.dropout = GVPDropout(dropout)


class NoisePredictionBlock(nn.Module):
    def __init__(self, in_scalar_dim, out_scalar_dim, vector_size, n_gvps=3, intermediate_scalar_dim=64):
        super().__init__()
        gvps = []
        for i in range(n_gvps):
            last = i == n_gvps - 1
            gvps.append(GVP(vector_size, 1 if last else vector_size, in_scalar_dim,
                            intermediate_scalar_dim if last else in_scalar_dim))
        self.gvps = nn.Sequential(*gvps)
        self.to_scalar_output = nn.Linear(intermediate_scalar_dim, out_scalar_dim)


class PharmRecGVP(nn.Module):
    pharmacophore_edges = [('pharm', 'ff', 'pharm'), ('prot', 'pf', 'pharm')]
    protein_edges = [('pharm', 'fp', 'prot'), ('prot', 'pp', 'prot')]
    all_edges = pharmacophore_edges + protein_edges

    def __init__(self, in_scalar_dim, in_vector_dim, out_scalar_dim, n_convs=4, n_message_gvps=3, n_update_gvps=2,
                 message_norm=10, n_noise_gvps=3, dropout=0.0):
        super().__init__()
        self.conv_layers = nn.ModuleList([
            GVPMultiEdgeConv(self.all_edges, in_scalar_dim, in_vector_dim, n_message_gvps, n_update_gvps,
                             message_norm=message_norm, dropout=dropout) for _ in range(n_convs)])
        self.noise_predictor = NoisePredictionBlock(in_scalar_dim, out_scalar_dim, in_vector_dim, n_noise_gvps)


class _DynamicsFn(torch.autograd.Function):
    """PharmRecDynamicsGVP.forward as one autograd node: forward = pf_train_forward (train-mode dropout inside the
    kernels), backward = pf_train_backward, which returns d(loss)/d(parameter) for every parameter as one flat vector.
    The inputs (x_t, h_t, t, coordinates) get no gradient: nothing upstream of them is trainable
    (pharmacodiff.py:162-243)."""

    @staticmethod
    def forward(ctx, mod, eng, x_t, h_t, t, prot_x, dropout, seed, flat_leaf):
        eps_h, eps_x = eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=dropout, seed=seed)
        mod._fwd_token += 1
        ctx.mod, ctx.eng, ctx.token = mod, eng, mod._fwd_token
        return eps_h, eps_x

    @staticmethod
    def backward(ctx, g_h, g_x):
        mod, eng = ctx.mod, ctx.eng
        if ctx.token != mod._fwd_token:
            raise RuntimeError("backward of a PharmRecDynamicsGVP call that is not the most recent training forward: "
                               "the engine keeps the activations of one forward at a time")
        g_h = torch.zeros(eng.Nf, eng.pharm_nf, device=eng.device) if g_h is None else g_h.contiguous()
        g_x = torch.zeros(eng.Nf, 3, device=eng.device) if g_x is None else g_x.contiguous()
        # ONE gradient for autograd: the flat vector, for the flat leaf that aliases every parameter (244 separate
        # gradient views and AccumulateGrad nodes cost 1.4 ms of host time per step -- more than the device could hide);
        # the hooks of the leaf re-bind each parameter's .grad to its slice (PharmRecDynamicsGVP._bind_grads)
        return (None,) * 8 + (eng.train_backward(g_h, g_x),)


class _LossFn(torch.autograd.Function):
    """PharmacophoreDiff.forward's loss (pharmacodiff.py:162-243, noise parameterisation) as one autograd node: forward =
    pf_train_loss_forward (COM removal, noising, train-mode dynamics, losses and metrics on the device), backward =
    pf_train_loss_backward_out with the upstream gradient of the output vector.  Output: [pos loss, feat loss, four metrics,
    total loss, total error, weighted total error]."""

    @staticmethod
    def forward(ctx, mod, eng, x0, h0, t_int, eps_x, eps_h, tabs, T, feat_norm, remove_com, weighted, dropout, seed, flat_leaf):
        out = eng.train_loss_forward(x0, h0, t_int, eps_x, eps_h, tabs[0], tabs[1], T, feat_norm, remove_com, weighted,
                                     dropout=dropout, seed=seed)
        mod._fwd_token += 1
        ctx.mod, ctx.eng, ctx.token = mod, eng, mod._fwd_token
        ctx.set_materialize_grads(False)
        # second output: the total loss (entry 6) as a 0-dim tensor of its own -- a step that differentiates only the total loss
        # then hands its upstream scalar straight to the C ABI instead of scattering it into a [9] vector first (a zero fill and
        # a copy launch per step)
        return out, out[6]

    @staticmethod
    def backward(ctx, g_out, g_total):
        mod, eng = ctx.mod, ctx.eng
        if ctx.token != mod._fwd_token:
            raise RuntimeError("backward of a loss forward that is not the most recent training forward: the engine keeps "
                               "the activations of one forward at a time")
        if g_out is None and g_total is None:
            return (None,) * 15
        leaf = mod.__dict__.get("_flat_leaf")
        # FlatAdam.zero_grad(lazy=True) left the clear to this backward: the gradient kernels STORE every element, so writing
        # into the bound flat gradient is the clear and the accumulation at once (no fill, no add: two 3 MB launches per step);
        # the parameters' .grad views stay bound, the accumulate hooks are not needed (no gradient is returned for the leaf)
        into = leaf.grad if (mod.__dict__.get("_grad_lazy_zero") and leaf is not None and leaf.grad is not None) else None
        if g_out is None:
            g = eng.train_loss_backward(g_total, g_total, out=into)          # total loss = pos loss + feat loss
        else:
            if g_total is not None:
                g_out = g_out.clone()
                g_out[6] += g_total
            g = eng.train_loss_backward_out(g_out, out=into)
        if into is not None:
            mod.__dict__["_grad_lazy_zero"] = False
            mod._last_flat_grad = into
            return (None,) * 15
        return (None,) * 14 + (g,)


class PharmRecDynamicsGVP(nn.Module):
    """Drop-in for pharmacoforge.models.dynamics_gvp.PharmRecDynamicsGVP (same constructor, same
    state-dict keys, same forward signature); forward runs on the MI355X through libpfdyn."""

    def __init__(self, n_pharm_scalars, n_prot_scalars, vector_size: int = 16, n_convs=4, n_hidden_scalars=128,
                 act_fn=nn.SiLU, message_norm=1, graph_cutoffs: dict = {}, n_message_gvps: int = 3,
                 n_update_gvps: int = 2, n_noise_gvps: int = 3, dropout: float = 0.0, ff_k: int = 0, pf_k: int = 0):
        super().__init__()
        self.graph_cutoffs = graph_cutoffs
        self.n_pharm_scalars, self.n_prot_scalars = n_pharm_scalars, n_prot_scalars
        self.vector_size, self.ff_k, self.pf_k = vector_size, ff_k, pf_k
        self._arch = dict(pharm_nf=n_pharm_scalars, rec_nf=n_prot_scalars, vector_size=vector_size,
                          n_hidden_scalars=n_hidden_scalars, n_convs=n_convs, n_message_gvps=n_message_gvps,
                          n_update_gvps=n_update_gvps, n_noise_gvps=n_noise_gvps, message_norm=message_norm,
                          ff_k=ff_k, pf_k=pf_k, graph_cutoffs=dict(graph_cutoffs))
        self.dropout_rate = dropout
        self.pharm_encoder = nn.Sequential(nn.Linear(n_pharm_scalars + 1, n_hidden_scalars), act_fn(),
                                           nn.LayerNorm(n_hidden_scalars))
        self.prot_encoder = nn.Sequential(nn.Linear(n_prot_scalars + 1, n_hidden_scalars), act_fn(),
                                          nn.LayerNorm(n_hidden_scalars))
        self.noise_predictor = PharmRecGVP(n_hidden_scalars, vector_size, n_pharm_scalars, n_convs, n_message_gvps,
                                           n_update_gvps, message_norm, n_noise_gvps, dropout)
        self._engine: Optional[PfEngine] = None
        self._weights_stamp = None
        self._batch_key = None
        self._bound_static = None                       # strong references to the tensors behind _batch_key
        self._flat: Optional[torch.Tensor] = None       # all parameters as one device vector (engine layout)
        self._flat_views = []                           # [(parameter, offset, numel)] in that layout
        self._fwd_token = 0

    # -- engine plumbing ------------------------------------------------------------------
    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def engine(self) -> PfEngine:
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("PharmRecDynamicsGVP runs only on an MI355X (move the model to 'cuda'); "
                               "there is no CPU fallback")
        if self._engine is None or self._engine.device != dev:
            self._engine = PfEngine(device=dev, **self._arch)
            self._engine.set_train_precision(getattr(self, "train_precision", "f32"))
            self._join_prefetch()
            self.__dict__["_twin"] = None           # (prefetch_graph makes a new one on this device)
            self._weights_stamp = None
            self._batch_key = None
            self._flat = None
        if self._flat is None or not self._views_intact():
            # first use on this device (or the parameters were re-allocated, e.g. by .to()): push the tensors through
            # the C ABI once, then re-home every parameter as a view of one flat device vector in the engine's
            # layout, so that an optimiser step is visible to the engine as a single device-to-device refresh
            self._engine.load_state_dict({k: v for k, v in self.state_dict().items()}, prefix="")
            self._flatten_parameters()
            self._weights_stamp = self._stamp()
        stamp = self._stamp()
        if stamp != self._weights_stamp:       # parameters were updated in place (optimiser step, load_state_dict)
            self._engine.set_flat_params(self._flat)
            self._weights_stamp = stamp
        return self._engine

    def set_train_precision(self, precision: str):
        """'f32' (default: what the reference trains in) or 'bf16' -- the labelled bf16 training leg (dense Linears of the message
        chains' forward and of the gradient kernels on bf16 matrix instructions, fp32 accumulation, fp32 master weights and
        optimiser).  No reference counterpart; inference is unaffected."""
        if str(precision).lower() not in ("f32", "fp32", "float32", "bf16", "bfloat16"):
            raise ValueError(f"train precision must be 'f32' or 'bf16', got {precision!r}")
        self.train_precision = "bf16" if str(precision).lower() in ("bf16", "bfloat16") else "f32"
        if self._engine is not None:
            self._engine.set_train_precision(self.train_precision)
        twin = self.__dict__.get("_twin")
        if twin is not None:
            self._join_prefetch()
            twin.set_train_precision(self.train_precision)

    def lane_engine(self, lane: int) -> PfEngine:
        """Engine of sampling lane ``lane``: lane 0 is engine(); further lanes are extra handles (own workspace) that carry
        the same weights -- PharmacophoreDiff.sample keeps several batches in flight on as many HIP streams, because most
        launches of a batched step leave part of the chip idle.  Call under the lane's stream."""
        eng0 = self.engine()
        if lane == 0:
            return eng0
        extra = self.__dict__.setdefault("_lane_engines", {})
        ent = extra.get(lane)
        if ent is None or ent[0].device != eng0.device:
            eng = PfEngine(device=eng0.device, **self._arch)
            eng.load_state_dict({k: v for k, v in self.state_dict().items()}, prefix="")
            ent = [eng, None]
            extra[lane] = ent
        if ent[1] != self._weights_stamp:
            ent[0].set_flat_params(self._flat)
            ent[1] = self._weights_stamp
        return ent[0]

    def _stamp(self):
        return tuple(p._version for p, _, _ in self._flat_views)

    def _views_intact(self) -> bool:
        # (an address inside the flat vector's allocation implies its device)
        base = self._flat.data_ptr()
        return all(p.data_ptr() == base + 4 * off for p, off, _ in self._flat_views)

    def _flatten_parameters(self):
        eng = self._engine
        layout = eng.param_layout()
        flat = torch.empty(eng.n_params, device=eng.device)
        named = dict(self.named_parameters())
        views = []
        for name, off, n in layout:
            p = named[name[len("dynamics."):]]
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            views.append((p, off, n))
        self._flat, self._flat_views = flat, views
        # the autograd face of the parameters: one leaf over the same storage (not a registered parameter: state_dict and
        # parameters() keep the reference's 244 tensors).  Its incoming-gradient hook notices gradients that were
        # cleared parameter by parameter (optimizer.zero_grad(), module.zero_grad()), its post-accumulate hook points
        # every parameter's .grad at its slice of the accumulated flat gradient.
        leaf = flat.detach().requires_grad_(True)
        leaf.register_hook(self._before_accumulate)
        leaf.register_post_accumulate_grad_hook(self._bind_grads)
        self.__dict__["_flat_leaf"] = leaf
        self._last_flat_grad = None
        # a gradient that reaches a parameter DIRECTLY (a regulariser on p, any loss term outside the fused nodes) accumulates
        # into p.grad -- a view of the flat gradient -- without passing the leaf: a deferred clear (FlatAdam.zero_grad(lazy=True))
        # has to happen in front of that too, whichever of the two arrives first in a backward
        for p, _, _ in views:
            if p.requires_grad and not getattr(p, "_pf_lazy_hook", False):
                p.register_hook(self._param_grad_arrives)
                p._pf_lazy_hook = True

    def _param_grad_arrives(self, grad):
        if self.__dict__.get("_grad_lazy_zero"):
            self.__dict__["_grad_lazy_zero"] = False
            leaf = self.__dict__.get("_flat_leaf")
            if leaf is not None and leaf.grad is not None:
                leaf.grad.zero_()
        return grad

    def _before_accumulate(self, grad):
        leaf = self.__dict__["_flat_leaf"]
        if self.__dict__.get("_grad_lazy_zero"):        # a deferred clear (FlatAdam.zero_grad(lazy=True)) met by a backward that accumulates
            self.__dict__["_grad_lazy_zero"] = False
            if leaf.grad is not None:
                leaf.grad.zero_()
        if leaf.grad is not None:
            p0 = next((p for p, _, _ in self._flat_views if p.requires_grad), None)
            if p0 is None or p0.grad is None or p0.grad.data_ptr() != leaf.grad.data_ptr() + 4 * next(o for q, o, _ in self._flat_views if q is p0):
                leaf.grad = None                # the per-parameter gradients were cleared or replaced: start afresh
        return grad

    def _bind_grads(self, leaf):
        g = leaf.grad
        self._last_flat_grad = g
        p0, o0 = next(((p, o) for p, o, _ in self._flat_views if p.requires_grad), (None, 0))
        if p0 is not None and p0.grad is not None and p0.grad.data_ptr() == g.data_ptr() + 4 * o0:
            return                              # accumulated in place: the views are still the gradient
        for p, off, n in self._flat_views:
            if p.requires_grad:
                p.grad = g[off:off + n].view(p.shape)

    def allreduce_gradients(self, group=None, average: bool = True):
        """Data-parallel training (one process per GPU): sum the gradients of all ranks with ONE all-reduce of the
        flat gradient vector (3.1 MB at dev.yml; RCCL when the tensors are on the GPU), then re-bind every
        parameter's .grad to its slice.  SURVEY.md 8(e)."""
        import torch.distributed as dist

        def reduce_(t):
            if t.is_cuda and dist.get_backend(group) != "nccl":      # gloo dry runs (ranks sharing a card): via the host
                c = t.cpu()
                dist.all_reduce(c, group=group)
                t.copy_(c)
            else:
                dist.all_reduce(t, group=group)
        params = [p for p, _, _ in self._flat_views] or [p for p in self.parameters() if p.numel() > 0]
        flat = getattr(self, "_last_flat_grad", None)
        off0 = self._flat_views[0][1] if self._flat_views else 0
        if flat is not None and params and params[0].grad is not None and params[0].grad.data_ptr() == flat.data_ptr() + 4 * off0:
            reduce_(flat)                        # the parameters' .grad are views of this vector already
            if average:
                flat /= dist.get_world_size(group)
            return flat
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
        reduce_(flat)
        if average:
            flat /= dist.get_world_size(group)
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        self._last_flat_grad = flat
        return flat

    def bind_graph(self, g: PocketGraph, prot_x: Optional[torch.Tensor] = None) -> PfEngine:
        """Upload the static part of a batch (pocket atoms, pp edges, graph boundaries) once.

        The upload is skipped only when the very same tensors are bound again: the key holds the storage address and
        version counter of every static tensor plus the ptr arrays' contents, and the module keeps a strong reference
        to those tensors while they are bound -- so neither CPython nor the caching allocator can hand their addresses
        to another batch (a temporary such as ``training_step(g.to(dev))`` is kept alive until the next bind)."""
        eng = self.engine()
        static, idx, key = self._bind_key(g, prot_x)
        if key != self._batch_key:
            pf = self.__dict__.get("_prefetch")
            if pf is not None:
                self._join_prefetch()
                if pf["key"] == key and prot_x is None:
                    # the batch is bound already -- on the twin handle, by prefetch_graph's worker while the previous step was being
                    # enqueued.  The twin becomes THE engine (everything else keeps reading self._engine); it takes the current
                    # weights with one device-side gather (the optimiser stepped on the other handle since the twin last ran)
                    self._engine, self._twin = self._twin, self._engine
                    eng = self._engine
                    eng.set_flat_params(self._flat)
                    self._batch_key = key
                    self._bound_static = pf["static"]
                    return eng
            eng.set_batch(static[0], g.prot_h, *idx, pocket_uid=g.pocket_uid if prot_x is None else None)
            self._batch_key = key
            self._bound_static = static
        return eng

    @staticmethod
    def _bind_key(g: PocketGraph, prot_x=None):
        static = (g.prot_x if prot_x is None else prot_x, g.prot_h, g.pp_src, g.pp_dst)
        idx = g.index_arrays_i32()
        key = (tuple((t.data_ptr(), t._version, tuple(t.shape), str(t.device)) for t in static),
               idx[0].tobytes(), idx[1].tobytes(),
               None if g.pocket_uid is None else tuple(g.pocket_uid.tolist()))
        return static, idx, key

    def prefetch_graph(self, g: PocketGraph) -> None:
        """Training loops that know their NEXT batch: bind it now, on a second handle (the "twin": own workspace, same weights),
        from a worker thread -- pf_set_pocket_batch is half a step's host time (the pass over 0.65 M pp edges at 256 pockets, the
        index tables, the staging copy: ~0.5 ms), and a training step costs the host as much as the device.  Call it once the
        current step's forward has been issued; the worker's C call runs (GIL released) while this thread enqueues the backward
        and the optimiser step, and the next bind_graph(g) adopts the twin instead of binding.  Both handles enqueue on the
        caller's stream, so the device sees the same order of work as without it; results are those of the single-handle loop
        bit for bit (tests/test_gpu_train.py).  Optional: without the call nothing changes."""
        import threading
        g = as_pocket_graph(g)
        self.engine()
        self._join_prefetch()
        static, idx, key = self._bind_key(g)
        if key == self._batch_key:
            return
        twin = self.__dict__.get("_twin")
        if twin is None or twin.device != self._engine.device:
            twin = PfEngine(device=self._engine.device, **self._arch)
            twin.load_state_dict({k: v for k, v in self.state_dict().items()}, prefix="")
            twin.set_train_precision(getattr(self, "train_precision", "f32"))
            self.__dict__["_twin"] = twin
        stream = torch.cuda.current_stream(twin.device)
        pf = {"key": key, "static": static, "err": None}

        def work():
            try:
                with torch.cuda.stream(stream):
                    twin.set_batch(static[0], g.prot_h, *idx, pocket_uid=g.pocket_uid)
            except BaseException as e:          # noqa: BLE001  (re-raised by the thread that joins)
                pf["err"] = e
        pf["thread"] = threading.Thread(target=work, name="pfdyn-prefetch", daemon=True)
        self.__dict__["_prefetch"] = pf
        pf["thread"].start()

    def _join_prefetch(self):
        pf = self.__dict__.get("_prefetch")
        if pf is None:
            return
        pf["thread"].join()
        self.__dict__["_prefetch"] = None
        if pf["err"] is not None:
            raise pf["err"]

    def forward(self, g, timestep: torch.Tensor, batch_idxs: Dict[str, torch.Tensor] = None):
        """(eps_h, eps_x) = dynamics(g, t): reads g.x_t, g.h_t (pharm) and g.prot_x, like the reference
        reads g.nodes['pharm'].data['x_t'/'h_t'] and g.nodes['prot'].data['x_0'] (dynamics_gvp.py:139-170)."""
        g = as_pocket_graph(g)
        return self.run(g, g.x_t, g.h_t, timestep, g.prot_x)

    def run(self, g: PocketGraph, x_t, h_t, timestep, prot_x):
        """The boundary call on explicit state tensors.  With autograd enabled and trainable parameters the result
        carries a graph whose backward runs the HIP gradient kernels; in train() mode GVPDropout is applied inside the
        kernels (its seed is drawn from torch's generator, so torch.manual_seed reproduces a step).  Otherwise
        (torch.no_grad() / frozen parameters) the pruned inference path runs."""
        eng = self.bind_graph(g)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p, _, _ in self._flat_views)
        if not need_grad and not (self.training and self.dropout_rate > 0):
            return eng.dynamics(x_t, h_t, timestep, prot_x=prot_x)
        p_drop = float(self.dropout_rate) if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0 else 0
        if not need_grad:
            return eng.train_forward(x_t, h_t, timestep, prot_x=prot_x, dropout=p_drop, seed=seed)
        return _DynamicsFn.apply(self, eng, x_t, h_t, timestep, prot_x, p_drop, seed, self.__dict__["_flat_leaf"])


class FlatAdam:
    """Adam over a PharmRecDynamicsGVP as ONE fused HIP kernel on the flat parameter vector (pf_adam_step) instead of
    a multi-tensor update over 244 tensors: same update rule as torch.optim.Adam(lr, betas, eps, weight_decay)
    (pharmacodiff.py:253).  Uses the flat gradient of the last backward (or of allreduce_gradients)."""

    def __init__(self, dynamics: "PharmRecDynamicsGVP", lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.dyn, self.lr, self.betas, self.eps, self.weight_decay = dynamics, lr, betas, eps, weight_decay
        self.t = 0
        self.exp_avg = self.exp_avg_sq = None
        self.param_groups = [{'lr': lr}]            # what LR schedulers / loggers look at

    def zero_grad(self, set_to_none: bool = False, lazy: bool = False):
        """Default (``set_to_none=False``): the flat gradient is zeroed IN PLACE -- one small kernel; the 244 parameters'
        ``.grad`` stay the views of it that the last backward bound, the next backward accumulates into it.  Clearing and
        re-creating 244 views per step costs ~0.8 ms of host time, which is what bounds a step that binds a new batch.
        ``set_to_none=True`` drops everything, like torch.optim's default.
        ``lazy=True`` (training loops that call zero_grad -> backward -> step): nothing is launched here; the next backward of
        the fused loss stores its gradient straight into the bound flat vector (clear and accumulation in one); any other
        gradient -- through the flat leaf or straight into a parameter (a regulariser on p) -- performs the deferred clear
        first, in whichever order a backward delivers them.  Until then ``.grad`` still shows the previous step's values."""
        views = self.dyn._flat_views
        leaf = self.dyn.__dict__.get("_flat_leaf")
        if not set_to_none and leaf is not None and leaf.grad is not None and views:
            p0, o0 = next(((p, o) for p, o, _ in views if p.requires_grad), (None, 0))
            if p0 is not None and p0.grad is not None and p0.grad.data_ptr() == leaf.grad.data_ptr() + 4 * o0:
                if lazy:
                    self.dyn.__dict__["_grad_lazy_zero"] = True
                    self.dyn._last_flat_grad = None         # step() without a backward in between has no gradient
                else:
                    self.dyn.__dict__["_grad_lazy_zero"] = False
                    leaf.grad.zero_()
                return
        self.dyn.__dict__["_grad_lazy_zero"] = False
        for p in ([p for p, _, _ in views] if views else self.dyn.parameters()):
            p.grad = None
        if leaf is not None:
            leaf.grad = None
        self.dyn._last_flat_grad = None

    # -- state: what torch.optim.Adam.state_dict() carries, on the flat vector (checkpoint / resume, rank-0 broadcast) --
    def ensure_state(self, t: int = None, lr: float = None):
        self.dyn.engine()
        if self.exp_avg is None:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.dyn._flat), torch.zeros_like(self.dyn._flat)
        if t is not None:
            self.t = int(t)
        if lr is not None:
            self.param_groups[0]['lr'] = float(lr)

    def state_dict_meta(self):
        return self.t, self.param_groups[0]['lr']

    def state_dict(self):
        self.ensure_state()
        return {'state': {'step': self.t, 'exp_avg': self.exp_avg.detach().cpu(), 'exp_avg_sq': self.exp_avg_sq.detach().cpu()},
                'param_groups': [{'lr': self.param_groups[0]['lr'], 'betas': self.betas, 'eps': self.eps,
                                  'weight_decay': self.weight_decay}],
                'layout': 'flat vector in pf_param_layout order'}

    def load_state_dict(self, sd):
        """Accepts this class's own state (flat moment vectors, marked by 'layout') and the per-parameter state of
        torch.optim.Adam -- what a checkpoint written by the reference's Lightning run holds (pharmacodiff.py:253-263:
        Adam(self.parameters())): parameter i of that optimiser is tensor i of the flat layout (state-dict order, the
        empty dummy_param included); parameters without an entry (never stepped) keep zero moments."""
        st, pg = sd['state'], sd['param_groups'][0]
        if 'layout' in sd:
            self.ensure_state(st['step'], pg['lr'])
            if st['exp_avg'].numel() != self.exp_avg.numel():
                raise ValueError("FlatAdam.load_state_dict: moment vectors do not match this model's parameter count")
            self.exp_avg.copy_(st['exp_avg'])
            self.exp_avg_sq.copy_(st['exp_avg_sq'])
        else:
            # parameter i of that optimiser = the i-th entry of the model's parameters(): PharmacophoreDiff's list is
            # gamma.gamma (a frozen nn.Parameter upstream, pharmacodiff.py:662-664: never stepped) followed by the dynamics
            # module's parameters in registration order, the empty dummy_param tensors included; an optimiser built on
            # model.dynamics.parameters() lacks the leading gamma
            where = {name: (off, n) for name, off, n in self.dyn.engine().param_layout()}
            names = ['dynamics.' + k for k, _ in self.dyn.named_parameters()]
            ids = list(pg.get('params', range(len(names))))
            if len(ids) == len(names) + 1:
                names = ['gamma.gamma'] + names
            elif len(ids) != len(names):
                raise ValueError(f"FlatAdam.load_state_dict: the optimiser state names {len(ids)} parameters, this model has "
                                 f"{len(names)} (+ gamma.gamma)")
            steps = [int(torch.as_tensor(st[i]['step']).item()) for i in ids if i in st and 'step' in st[i]]
            self.ensure_state(max(steps) if steps else 0, pg['lr'])
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()
            for i, name in zip(ids, names):
                if i not in st or name not in where or where[name][1] == 0:
                    continue
                off, n = where[name]
                for key, dst in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq)):
                    src = st[i][key].reshape(-1)
                    if src.numel() != n:
                        raise ValueError(f"FlatAdam.load_state_dict: moment of {name} has {src.numel()} elements, expected {n}")
                    dst[off:off + n].copy_(src)
        self.betas, self.eps = tuple(pg.get('betas', self.betas)), pg.get('eps', self.eps)
        self.weight_decay = pg.get('weight_decay', self.weight_decay)

    def step(self):
        dyn = self.dyn
        g = getattr(dyn, "_last_flat_grad", None)
        if g is None:
            raise RuntimeError("FlatAdam.step(): no gradient (run loss.backward() on a training forward first)")
        # the engine that produced this gradient (the forward's bind checked the parameter views and synchronised it; another
        # pass over the 244 parameters here is 0.1 ms of a 1.6 ms step); engine() only when there is none yet
        # Cheap guard on that fast path: parameters re-materialised between backward() and step() (module.to(),
        # an assignment to p.data) no longer view the flat vector; the first and the last view are checked every step,
        # all of them every 64th, and a mismatch falls back to engine(), which re-homes and re-synchronises them.
        fast = dyn._engine is not None and dyn._flat is not None and dyn._flat.is_cuda and bool(dyn._flat_views)
        if fast:
            base = dyn._flat.data_ptr()
            ends = (dyn._flat_views[0], dyn._flat_views[-1])
            fast = all(p.data_ptr() == base + 4 * off for p, off, _ in ends) and (self.t % 64 != 63 or dyn._views_intact())
        if not fast:
            eng = dyn.engine()
            if g.numel() != dyn._flat.numel() or g.device != dyn._flat.device:
                raise RuntimeError("FlatAdam.step(): the parameters were re-allocated after backward(); run the step's "
                                   "forward and backward again")
        else:
            eng = dyn._engine
        if self.exp_avg is None:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(dyn._flat), torch.zeros_like(dyn._flat)
        self.t += 1
        eng.adam_step(dyn._flat, g, self.exp_avg, self.exp_avg_sq, self.t, self.param_groups[0]['lr'], self.betas, self.eps,
                      self.weight_decay)
        # the engine already holds the new values (its kernel wrote the flat vector the parameter views share) and the kernel does not
        # touch the views' version counters: the stamp engine() took for this step still describes them.  (Re-reading the 244
        # counters here was 40 us of host time per step; a parameter modified in place between the forward and this call makes the
        # next engine() upload the flat vector once more, which is correct either way.)


class PharmSizeDistribution:
    """pharmacoforge/models/n_nodes_dist.py:7-14 (sample_uniformly only; sample_variety is broken upstream)."""

    def __init__(self, dataset_dir=None):
        pass

    def sample_uniformly(self, n_replicates) -> torch.Tensor:
        return torch.from_numpy(np.random.randint(3, 9, n_replicates))


# ---------------------------------------------------------------------------------------------
class PharmacophoreDiff(_Base):
    """Drop-in for pharmacoforge.models.pharmacodiff.PharmacophoreDiff on the sampling / evaluation
    path.  Constructor arguments, attribute names, state-dict keys ('gamma.gamma', 'dynamics.*') and
    the metric names follow the reference (pharmacodiff.py:27-78, 231-239)."""

    def __init__(self, pharm_nf, rec_nf, ph_type_map: List[str], processed_data_dir: Path = None,
                 n_timesteps: int = 1000, graph_config={}, dynamics_config={}, lr_scheduler_config={},
                 sample_interval: float = 1, val_loss_interval: float = 1, batch_size: int = 64,
                 pharms_per_pocket: int = 8, n_pockets_to_sample: int = 8, precision=1e-4,
                 pharm_feat_norm_constant=1, endpoint_param_feat: bool = False, endpoint_param_coord: bool = False,
                 weighted_loss: bool = False, remove_com: bool = True, **kwargs):
        super().__init__()
        self.n_pharm_feats, self.n_prot_feats = pharm_nf, rec_nf
        self.batch_size = batch_size
        self.ph_type_map = ph_type_map
        self.n_timesteps = n_timesteps
        self.remove_com = remove_com
        self.pharm_feat_norm_constant = pharm_feat_norm_constant
        self.endpoint_param_feat, self.endpoint_param_coord = endpoint_param_feat, endpoint_param_coord
        self.weighted_loss = weighted_loss
        self.pharm_size_dist = PharmSizeDistribution(processed_data_dir)
        self.gamma = PredefinedNoiseSchedule(noise_schedule='polynomial_2', timesteps=n_timesteps, precision=precision)
        self.dynamics = PharmRecDynamicsGVP(pharm_nf, rec_nf, **graph_config, **dynamics_config)
        self.lr_scheduler_config = lr_scheduler_config
        self.sample_interval, self.pharms_per_pocket = sample_interval, pharms_per_pocket
        self.n_pockets_to_sample = n_pockets_to_sample
        self.last_sample_marker = 0
        self.last_epoch_exact = 0
        self.val_loss_interval = val_loss_interval
        self._hparams_dict = dict(pharm_nf=pharm_nf, rec_nf=rec_nf, ph_type_map=ph_type_map,
                                  processed_data_dir=processed_data_dir, n_timesteps=n_timesteps,
                                  graph_config=graph_config, dynamics_config=dynamics_config,
                                  lr_scheduler_config=lr_scheduler_config, sample_interval=sample_interval,
                                  val_loss_interval=val_loss_interval, batch_size=batch_size,
                                  pharms_per_pocket=pharms_per_pocket, n_pockets_to_sample=n_pockets_to_sample,
                                  precision=precision, pharm_feat_norm_constant=pharm_feat_norm_constant,
                                  endpoint_param_feat=endpoint_param_feat, endpoint_param_coord=endpoint_param_coord,
                                  weighted_loss=weighted_loss, remove_com=remove_com, **kwargs)
        if pl is not None:
            self.save_hyperparameters()
        self._coef = None

    # -- Lightning-free conveniences ----------------------------------------------------------
    if pl is None:
        @property
        def device(self) -> torch.device:
            return next(self.parameters()).device

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location="cpu", **overrides):
            """A Lightning .ckpt is a torch-loadable dict with 'state_dict' and 'hyper_parameters'."""
            ckpt = torch.load(str(checkpoint_path), map_location=map_location, weights_only=False)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(overrides)
            model = cls(**hp)
            model.load_state_dict(ckpt["state_dict"], strict=True)
            return model

        def log_dict(self, *a, **k):
            pass

    def save_checkpoint(self, path, **extra):
        """Write a file ``load_from_checkpoint`` (ours or Lightning's) can read.  ``extra``: further top-level entries in
        Lightning's checkpoint vocabulary ('epoch', 'global_step', 'optimizer_states', 'lr_schedulers')."""
        torch.save({"state_dict": self.state_dict(), "hyper_parameters": self._hparams_dict, **extra}, str(path))

    # -- small algebra (pharmacodiff.py:80-86, 140-160) ----------------------------------------
    def sigma(self, gamma):
        return _sigma(gamma)

    def alpha(self, gamma):
        return _alpha(gamma)

    def sigma_and_alpha_t_given_s(self, gamma_t, gamma_s):
        return sigma_and_alpha_t_given_s(gamma_t, gamma_s)

    def step_coefficients(self):
        if self._coef is None:
            self._coef = step_coefficients(self.gamma.gamma, self.n_timesteps)
        return self._coef

    # -- sampling (pharmacodiff.py:433-514) ------------------------------------------------------
    @torch.no_grad()
    def sample_given_receptor(self, g, init_pharm_com: torch.Tensor = None, visualize_trajectory: bool = False,
                              noise: torch.Tensor = None) -> List[SampledPharmacophore]:
        """Reverse diffusion for a batch of pocket graphs; the whole T-step loop is enqueued on the
        current HIP stream by pf_sample (no host synchronisation inside the loop).

        ``noise`` ([T+1, Nf, 3+pharm_nf], optional) injects the Gaussian draws (initial draw first; x
        columns before h columns -- the reference's draw order, pharmacodiff.py:455-456, 423-424): replaying a
        reference run means passing its draws here.  When omitted all T+1 draws come from ONE torch.randn call on the
        model's device (same distribution, reproducible under torch.manual_seed; the reference's 2(T+1) separate calls
        would cost a thousand launches per batch and cannot reproduce a CUDA generator's stream on ROCm anyway)."""
        return self._sample_finish(self._sample_fetch(self._sample_enqueue(g, init_pharm_com, visualize_trajectory, noise)))

    def _sample_enqueue(self, g, init_pharm_com=None, visualize_trajectory=False, noise=None, lane: int = 0):
        """First half of sample_given_receptor: upload the batch and enqueue the whole reverse process (asynchronous) on the
        current stream; ``lane`` > 0: on that lane's own handle (PharmRecDynamicsGVP.lane_engine)."""
        g = as_pocket_graph(g)
        dev = self.device
        T, Nf, nf = self.n_timesteps, g.num_nodes("pharm"), self.n_pharm_feats
        if noise is None:
            noise = torch.randn(T + 1, Nf, 3 + nf, device=dev)
        if lane == 0:
            eng = self.dynamics.bind_graph(g)
        else:
            eng = self.dynamics.lane_engine(lane)
            eng.set_batch(g.prot_x, g.prot_h, g.prot_ptr, g.pharm_ptr, g.pp_src, g.pp_dst, pocket_uid=g.pocket_uid)
        coef = self.step_coefficients()
        if getattr(self, "_coef_arr", None) is None or self._coef_arr[0] != T:      # 500-1000 ctypes structs: built once
            self._coef_arr = (T, eng.coef_array(coef, reversed(range(T))))
        arr = self._coef_arr[1]
        com = None
        if init_pharm_com is not None:
            com = init_pharm_com if init_pharm_com.is_cuda else init_pharm_com.float().pin_memory().to(dev, non_blocking=True)
        res = eng.sample(arr, T, noise if noise.is_cuda else noise.float().pin_memory().to(dev, non_blocking=True),
                         init_pharm_com=com, ep_coord=self.endpoint_param_coord,
                         ep_feat=self.endpoint_param_feat, feat_norm_constant=float(self.pharm_feat_norm_constant),
                         trajectory=visualize_trajectory)
        # results go to pinned host memory with copies enqueued right behind the batch's kernels, and an event marks their
        # completion: whatever is enqueued afterwards (the next batch) does not delay the fetch of this one
        host = tuple(None if r is None else torch.empty(r.shape, dtype=r.dtype, pin_memory=True).copy_(r, non_blocking=True)
                     for r in res)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        return g, host, visualize_trajectory, done, eng

    def _sample_fetch(self, pending):
        """Results of an enqueued batch on the host (waits for that batch only).  The run's validity word arrived with them
        (pf_sample_end): an exchange time-out inside a merged launch raises PfError HERE, before anything is built from x_0 / h_0."""
        g, host, traj, done, eng = pending
        done.synchronize()
        eng.sample_status()
        return g, host, traj

    def _sample_finish(self, pending) -> List[SampledPharmacophore]:
        """Second half: per-graph SampledPharmacophores (host work only once the results are fetched)."""
        g, res, visualize_trajectory = pending[:3]
        x0, h0 = res[0].cpu(), res[1].cpu()
        traj_x = res[2].cpu() if visualize_trajectory else None
        traj_h = res[3].cpu() if visualize_trajectory else None
        out: List[SampledPharmacophore] = []
        g_cpu = g.to("cpu")
        for gi, g_i in enumerate(unbatch(g_cpu)):
            f0, f1 = int(g.pharm_ptr[gi]), int(g.pharm_ptr[gi + 1])
            g_i.pharm_x0, g_i.pharm_h0 = x0[f0:f1], h0[f0:f1]
            kwargs = {"g": g_i, "pharm_type_map": self.ph_type_map}
            if visualize_trajectory:
                kwargs["traj_frames"] = (traj_x[:, f0:f1], traj_h[:, f0:f1])
            out.append(SampledPharmacophore(**kwargs))
        return out

    def sample(self, ref_graphs: List[PocketGraph], n_pharms: List[List[int]], max_batch_size: int = 32,
               init_pharm_com: torch.Tensor = None, visualize_trajectory: bool = False,
               rank: int = 0, world_size: int = 1, noise=None, lanes: int = None) -> List[List[SampledPharmacophore]]:
        """pharmacodiff.py:516-578: one pocket copy per requested pharmacophore, flattened in pocket order and cut into
        batches of ``max_batch_size`` in list order.

        With ``world_size`` > 1 whole BATCHES are dealt over the ranks (greedy by protein-protein edge count,
        sharding.shard_by_work; no data-path collective): a batch holds the same graphs, in the same order, whatever the
        number of GPUs, so every sample is bitwise the one a single GPU produces.  Each rank returns only its own
        samples, still grouped per pocket (shorter or empty lists for the others).

        ``noise``: None -- batch i draws its [T+1, Nf_i, 3+nf] Gaussians from a device generator seeded with
        (one draw from torch's global CPU generator) + i, i.e. reproducible under torch.manual_seed and independent of
        rank / world_size when the ranks share the seed; or a list with one such tensor per batch (replaying a
        reference run: tests/golden/sample_multi.npz).

        ``lanes`` (default ``self.sample_lanes``; None = 2, or 4 when the batches hold at most 32 graphs): batches in flight at
        once, each on its own HIP stream and handle.  Four of
        the five launches of a batched step occupy part of the chip, so independent batches overlap (2 lanes: +16 % at
        batches of 128, +34 % at batches of 32; 4 lanes: +27 % / +58 %); a batch's result does not depend on the lane it ran on."""
        from .sharding import shard_by_work
        ref_graphs = [as_pocket_graph(g) for g in ref_graphs]
        n_receptors = len(ref_graphs)
        if n_receptors == 0:
            return []
        if init_pharm_com is None:
            init_pharm_com = torch.stack([g.prot_x.mean(dim=0) for g in ref_graphs], dim=0)
        # the requested graphs in pocket order: (pocket, number of centers); the copies themselves are made batch by batch
        # inside the pipeline (30,000 graph objects up front cost a dataset-scale run most of a second before the device starts)
        graph_ref_idx = [rec_idx for rec_idx in range(n_receptors) for _ in n_pharms[rec_idx]]
        graph_size = [int(n) for rec_idx in range(n_receptors) for n in n_pharms[rec_idx]]
        n_graphs = len(graph_ref_idx)

        def make_batch(idx):
            gs, a = [], 0
            while a < len(idx):                         # runs of one pocket: one copy_graph call each (its copies share a pocket_uid)
                b = a
                while b < len(idx) and graph_ref_idx[idx[b]] == graph_ref_idx[idx[a]]:
                    b += 1
                gs.extend(copy_graph(ref_graphs[graph_ref_idx[idx[a]]], n_copies=b - a,
                                     pharm_feats_per_copy=torch.tensor([graph_size[i] for i in idx[a:b]])))
                a = b
            return batch_graphs(gs)
        batches = [list(range(s0, min(s0 + max_batch_size, n_graphs))) for s0 in range(0, n_graphs, max_batch_size)]
        if world_size > 1:
            epp = [int(g.pp_src.numel()) for g in ref_graphs]
            work = [sum(epp[graph_ref_idx[i]] + 1 for i in idx) for idx in batches]
            mine = shard_by_work(work, world_size)[rank]
        else:
            mine = list(range(len(batches)))
        T, nf = self.n_timesteps, self.n_pharm_feats
        base_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if noise is None else 0
        # every graph's initial center of mass on the device with ONE copy before the pipeline starts (a batch is a
        # contiguous range of graphs); per-batch host -> device copies would each wait for the batch in flight
        coms_dev = init_pharm_com[graph_ref_idx].to(self.device, torch.float32) if n_graphs else None
        sampled = {}
        # Pipeline over the batches: the host work of a batch (collating the next one, splitting the previous one's results
        # into SampledPharmacophores) runs while the device works on another batch.  Nothing in it waits for the device
        # except the fetch of a finished batch: uploads go through pinned memory, pf_set_pocket_batch is asynchronous, and
        # the one workspace of the handle is reused in stream order.  The batched graph stays on the host: the engine
        # uploads what it needs once, and the per-graph views are host tensors anyway.
        def finish(done):
            for i, p in zip(done[0], self._sample_finish(done[1])):
                sampled[i] = p
        want = self.sample_lanes if lanes is None else lanes
        if want is None:                                # batches of up to 32 graphs leave more of the chip idle: four lanes (811 k against
            want = 4 if max((len(batches[bi]) for bi in mine), default=0) <= 32 else 2      # 737 k sample-steps/s with two); larger ones: two
        n_lanes = max(1, min(int(want), len(mine)))
        cur = torch.cuda.current_stream(self.device)
        streams = [cur] + [torch.cuda.Stream(device=self.device) for _ in range(n_lanes - 1)]
        for st in streams[1:]:
            st.wait_stream(cur)                         # coms_dev (and the caller's earlier work) are ready
        pending = []                                    # enqueued batches, oldest first: at most n_lanes
        for k, bi in enumerate(mine):
            idx = batches[bi]
            batch_g = make_batch(idx)
            init_coms = coms_dev[idx[0]:idx[-1] + 1]
            lane = k % n_lanes
            with torch.cuda.stream(streams[lane]):
                if noise is None:
                    gen = torch.Generator(device=self.device).manual_seed(base_seed + bi)
                    nz = torch.randn(T + 1, batch_g.num_nodes("pharm"), 3 + nf, device=self.device, generator=gen)
                else:
                    nz = noise[bi]
                # the bind is asynchronous (stream-ordered behind the lane's previous batch), nothing here waits for the device
                enq = self._sample_enqueue(batch_g, init_coms, visualize_trajectory, nz, lane=lane)
            pending.append((idx, enq))
            if len(pending) > n_lanes:                  # every lane has a batch queued behind the one it runs: fetch the oldest
                i0, e0 = pending.pop(0)                 # (its lane goes straight into the batch just enqueued) and split it on
                finish((i0, self._sample_fetch(e0)))    # the host while the device works
        for i0, e0 in pending:
            finish((i0, self._sample_fetch(e0)))
        for st in streams[1:]:
            cur.wait_stream(st)
        per_pocket, end = [], 0
        for rec_idx in range(n_receptors):
            start, end = end, end + len(n_pharms[rec_idx])
            per_pocket.append([sampled[i] for i in range(start, end) if i in sampled])
        return per_pocket

    # -- training-loss forward (pharmacodiff.py:162-243), evaluation only ---------------------------
    def _check_injected_t(self, t_int, B):
        """A caller-supplied t_int indexes tables of n_timesteps + 1 entries on the device: refuse anything else here (the
        values the module draws itself, randint(0, T), need no check and no synchronisation)."""
        if t_int.numel() != B:
            raise ValueError(f"t_int has {t_int.numel()} entries for a batch of {B} graphs")
        lo, hi = int(t_int.min()), int(t_int.max())
        if lo < 0 or hi > self.n_timesteps:
            raise ValueError(f"t_int must lie in [0, {self.n_timesteps}], got [{lo}, {hi}]")

    def forward(self, g, phase: str = 'train', t_int: torch.Tensor = None, eps: Dict[str, torch.Tensor] = None):
        """Losses and metrics of one batch (pharmacodiff.py:162-243) with the dynamics -- and, when autograd is on,
        its backward -- evaluated by the HIP kernels.  ``t_int`` / ``eps`` inject the random draws."""
        g = as_pocket_graph(g)
        dev = self.device
        self.__dict__["_fused_sums"] = None
        if self.fused_loss and dev.type == "cuda" and not self.endpoint_param_feat and not self.endpoint_param_coord:
            return self._forward_fused(g, phase, t_int, eps)
        bidx = get_batch_idxs(g)
        bp, br = bidx['pharm'].to(dev), bidx['prot'].to(dev)
        B = g.batch_size
        # the (host) ptr array on the device, cached on the graph object like the batch indices: a pageable host -> device
        # copy waits for everything enqueued before it, i.e. for the previous training step
        ck = g.__dict__.get("_ptrf_cache")
        if ck is None or ck[0] != (str(dev), g.pharm_ptr.data_ptr(), int(g.pharm_ptr[-1])):
            ck = ((str(dev), g.pharm_ptr.data_ptr(), int(g.pharm_ptr[-1])), g.pharm_ptr.to(dev))
            g.__dict__["_ptrf_cache"] = ck
        ptr_f = ck[1]

        def seg_mean(x):
            out = torch.zeros(B, 3, device=dev).index_add_(0, bp, x)
            return out / (ptr_f[1:] - ptr_f[:-1]).clamp(min=1).unsqueeze(1)

        h0 = g.pharm_h0.to(dev) / self.pharm_feat_norm_constant
        com = seg_mean(g.pharm_x0.to(dev))
        x0 = g.pharm_x0.to(dev) - com[bp]
        prot_x = g.prot_x.to(dev) - com[br]
        if t_int is None:
            t_int = torch.randint(0, self.n_timesteps, size=(B,), device=dev)
        else:
            self._check_injected_t(t_int, B)
        t = t_int.to(dev).float() / self.n_timesteps
        if eps is None:
            eps = {'h': torch.randn(h0.shape, device=dev), 'x': torch.randn(x0.shape, device=dev)}
        eps = {k: v.to(dev) for k, v in eps.items()}
        gamma_t = self.gamma(t)
        alpha_t = self.alpha(gamma_t)[bp][:, None]
        sigma_t = self.sigma(gamma_t)[bp][:, None]
        x_t = alpha_t * x0 + sigma_t * eps['x']
        h_t = alpha_t * h0 + sigma_t * eps['h']
        if self.remove_com:
            c = seg_mean(x_t)
            x_t = x_t - c[bp]
            prot_x = prot_x - c[br]
            sampled_com = c[bp]
        h_dyn, x_dyn = self.dynamics.run(g, x_t, h_t, t, prot_x)
        if self.endpoint_param_feat:
            h_0_pred = h_dyn
            h_loss = F.cross_entropy(h_0_pred, h0.argmax(dim=1), reduction='none')
        else:
            h_loss = (eps['h'] - h_dyn).square().sum(dim=1)
            h_0_pred = (h_t - sigma_t * h_dyn) / alpha_t
        if self.endpoint_param_coord:
            if self.remove_com:
                x_dyn = x_dyn + sampled_com
            x_0_pred = x_dyn
            x_loss = (x_0_pred - x0).square().sum(dim=1)
        else:
            x_loss = (eps['x'] - x_dyn).square().sum(dim=1)
            x_0_pred = (x_t - sigma_t * x_dyn) / alpha_t
        weight_metric = 1 - t[bp]
        weight_loss = weight_metric if self.weighted_loss else torch.ones_like(weight_metric)
        losses = {phase + ' pos loss': (x_loss * weight_loss).sum() / eps['x'].numel(),
                  phase + ' feat loss': (h_loss * weight_loss).sum() / eps['h'].numel()}
        with torch.no_grad():
            err = (x_0_pred - x0).square().sum(dim=1)
            hit = (h_0_pred.argmax(dim=1) == h0.argmax(dim=1)).float()
            metrics = {phase + ' position error': err.mean(), phase + ' weighted position error': (weight_metric * err).mean(),
                       phase + ' accuracy': hit.mean(), phase + ' weighted accuracy': (weight_metric * hit).mean()}
        return losses, metrics

    sample_lanes = None    # batches that sample() keeps in flight at once (HIP streams / handles); None: 2, or 4 for batches of <= 32
                           # graphs; 1: strictly one after the other
    fused_loss = True      # noise-parameterised losses run as one C-ABI call (pf_train_loss_forward); False: the framework-op restatement above

    def _loss_tables(self):
        """alpha(gamma(k / T)), sigma(gamma(k / T)) for k = 0..T with the reference's own fp32 expressions (pharmacodiff.py:186-190,
        582-668), on the device: what forward() looks up at t_int."""
        tabs = self.__dict__.get("_loss_tabs")
        if tabs is None or tabs[0].device != self.device:
            t = torch.arange(0, self.n_timesteps + 1, device=self.device).float() / self.n_timesteps
            gamma_t = self.gamma(t)
            tabs = (self.alpha(gamma_t).float().contiguous(), self.sigma(gamma_t).float().contiguous())
            self.__dict__["_loss_tabs"] = tabs
        return tabs

    def _forward_fused(self, g, phase, t_int, eps):
        """forward() for the noise parameterisation through pf_train_loss_forward: same draws (t_int, eps), same losses and
        metrics; the pocket's coordinates are the ones bound to the engine (bind_graph), nothing else is uploaded but the
        clean centers."""
        dev = self.device
        dyn = self.dynamics
        eng = dyn.bind_graph(g)
        B = g.batch_size
        x0, h0 = g.pharm_x0.to(dev), g.pharm_h0.to(dev)
        if t_int is None:
            t_int = torch.randint(0, self.n_timesteps, size=(B,), device=dev, dtype=torch.int32)    # (the form the C ABI takes)
        else:
            self._check_injected_t(t_int, B)
        if eps is None:
            eps = {'h': torch.randn(h0.shape, device=dev), 'x': torch.randn(x0.shape, device=dev)}
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p, _, _ in dyn._flat_views)
        p_drop = float(dyn.dropout_rate) if dyn.training else 0.0
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0 else 0
        args = (x0, h0, t_int, eps['x'], eps['h'], self._loss_tables(), self.n_timesteps, float(self.pharm_feat_norm_constant),
                bool(self.remove_com), bool(self.weighted_loss), p_drop, seed)
        if need_grad:
            out, total = _LossFn.apply(dyn, eng, *args, dyn.__dict__["_flat_leaf"])
        else:
            out = eng.train_loss_forward(args[0], args[1], args[2], args[3], args[4], args[5][0], args[5][1], *args[6:10],
                                         dropout=p_drop, seed=seed)
            total = out[6]
        losses = {phase + ' pos loss': out[0], phase + ' feat loss': out[1]}
        m = out.detach()
        metrics = {phase + ' position error': m[2], phase + ' weighted position error': m[3],
                   phase + ' accuracy': m[4], phase + ' weighted accuracy': m[5]}
        # the sums a step derives from these (total loss, total error, weighted total error) came out of the same kernel:
        # training_step / validation_step pick them up instead of spending framework launches on three additions
        self.__dict__["_fused_sums"] = (total, m[7], m[8])
        return losses, metrics

    def configure_optimizers(self):
        """pharmacodiff.py:253-263: Adam + ReduceLROnPlateau from lr_scheduler_config."""
        cfg = self.lr_scheduler_config
        optimizer = torch.optim.Adam(self.parameters(), lr=cfg.get('base_lr', 1e-4), weight_decay=cfg.get('weight_decay', 0.0))
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, **cfg.get('reducelronplateau', {}))
        return {'optimizer': optimizer,
                'lr_scheduler': {"scheduler": scheduler, "monitor": cfg.get('monitor', 'val total loss'),
                                 "interval": cfg.get('interval', 'step'), "frequency": cfg.get('frequency', 1)}}

    # -- trainer context: Lightning's when attached, else a two-attribute stand-in (attach_trainer) -----------------
    def attach_trainer(self, datamodule, train_dataloader=None, current_epoch: int = 0, optimizer=None):
        """For Lightning-free drivers (train.py here): gives training_step what it reads from ``self.trainer`` in the
        reference -- ``trainer.datamodule.val_dataset`` (pharmacodiff.py:323), ``len(trainer.train_dataloader)``
        (:245) -- plus ``current_epoch`` and the optimiser whose learning rate is logged (:287-290)."""
        from types import SimpleNamespace
        self.__dict__["_pf_trainer"] = SimpleNamespace(datamodule=datamodule, train_dataloader=train_dataloader,
                                                       current_epoch=current_epoch, optimizer=optimizer)

    def _trainer_ctx(self):
        tr = self.__dict__.get("_pf_trainer")
        if tr is not None:
            return tr
        if pl is not None:
            try:
                tr = self.trainer
                from types import SimpleNamespace
                opt = None
                try:
                    opt = self.optimizers()
                except Exception:
                    pass
                return SimpleNamespace(datamodule=tr.datamodule, train_dataloader=tr.train_dataloader,
                                       current_epoch=self.current_epoch, optimizer=opt)
            except Exception:
                return None
        return None

    def num_training_batches(self):
        return len(self._trainer_ctx().train_dataloader)                        # pharmacodiff.py:244-245

    def training_step(self, batch, batch_idx, t_int: torch.Tensor = None, eps: Dict[str, torch.Tensor] = None):
        """pharmacodiff.py:265-296: total loss = pos loss + feat loss (the caller -- Lightning or a plain loop --
        runs .backward() and the optimiser step).  With a trainer context (Lightning's, or attach_trainer) the
        fractional epoch is tracked and, every ``sample_interval`` epochs, sample_and_analyze runs inside the step
        and its metrics join the logged ones (:281-284); without one (a bare loop over tensors) that part is skipped."""
        phase = 'train'
        tr = self._trainer_ctx()
        ph_quality_metrics, epoch_exact = {}, None
        if tr is not None and tr.train_dataloader is not None:
            epoch_exact = tr.current_epoch + batch_idx / max(len(tr.train_dataloader), 1)       # :270
            if epoch_exact - self.last_sample_marker >= self.sample_interval and tr.datamodule is not None:   # :281
                # sampled BEFORE this step's forward (the reference samples between its forward and the caller's
                # backward, :273-284): the engine keeps the activations of one training forward at a time, and sampling
                # rebinds its workspace; the losses / metrics do not depend on the order, only the RNG stream does
                ph_quality_metrics = self.sample_and_analyze()
                self.last_sample_marker = epoch_exact
        loss_dict, metrics_dict = self.forward(batch, phase=phase, t_int=t_int, eps=eps)
        sums = self.__dict__.pop("_fused_sums", None)
        if sums is not None:                       # (the fused loss path computed them with the losses)
            loss_dict[phase + ' total loss'], metrics_dict[phase + ' total error'], metrics_dict[phase + ' weighted total error'] = sums
        else:
            loss_dict[phase + ' total loss'] = torch.sum(torch.stack(list(loss_dict.values()), dim=0))
            metrics_dict[phase + ' total error'] = metrics_dict[phase + ' position error'] + 1 - metrics_dict[phase + ' accuracy']
            metrics_dict[phase + ' weighted total error'] = (metrics_dict[phase + ' weighted position error'] + 1
                                                             - metrics_dict[phase + ' weighted accuracy'])
        metrics_dict.update(ph_quality_metrics)
        if epoch_exact is not None:
            if tr.optimizer is not None:
                metrics_dict['lr'] = tr.optimizer.param_groups[0]['lr']                          # :287-290
            loss_dict['epoch_exact'] = epoch_exact
            self.last_epoch_exact = epoch_exact
        bs = as_pocket_graph(batch).batch_size
        self.log_dict(loss_dict, on_step=True, on_epoch=True, prog_bar=True, logger=True, batch_size=bs)
        self.log_dict(metrics_dict, on_step=True, on_epoch=True, prog_bar=True, logger=True, batch_size=bs)
        self.last_metrics = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in {**loss_dict, **metrics_dict}.items()}
        return loss_dict[phase + ' total loss']

    @torch.no_grad()
    def sample_and_analyze(self, rank: int = 0, world_size: int = 1, process_group=None):
        """pharmacodiff.py:320-357: ``n_pockets_to_sample`` random validation pockets (with replacement, like the
        reference's randint), ``pharms_per_pocket`` pharmacophores each with the reference pharmacophore's size and
        center of mass, batches of 64, validity of the lot."""
        val_dataset = self._trainer_ctx().datamodule.val_dataset
        pocket_idxs = torch.randint(low=0, high=len(val_dataset), size=(self.n_pockets_to_sample,))
        pockets = [val_dataset[int(i)] for i in pocket_idxs]
        return self.sample_and_analyze_graphs(pockets, rank=rank, world_size=world_size, process_group=process_group)

    def validation_step(self, batch, batch_idx):
        phase = 'val'
        loss_dict, metrics_dict = self.forward(batch, phase=phase)
        sums = self.__dict__.pop("_fused_sums", None)
        if sums is not None:
            loss_dict[phase + ' total loss'], metrics_dict[phase + ' total error'], metrics_dict[phase + ' weighted total error'] = sums
        else:
            loss_dict[phase + ' total loss'] = sum(list(loss_dict.values()))
            metrics_dict[phase + ' total error'] = metrics_dict[phase + ' position error'] + 1 - metrics_dict[phase + ' accuracy']
            metrics_dict[phase + ' weighted total error'] = (metrics_dict[phase + ' weighted position error'] + 1
                                                             - metrics_dict[phase + ' weighted accuracy'])
        loss_dict['epoch_exact'] = self.last_epoch_exact
        self.log_dict(loss_dict, on_step=False, on_epoch=True, prog_bar=True, logger=True, batch_size=batch.batch_size)
        self.log_dict(metrics_dict, on_step=False, on_epoch=True, prog_bar=True, logger=True, batch_size=batch.batch_size)
        return loss_dict[phase + ' total loss']

    def sample_and_analyze_graphs(self, pockets: List[PocketGraph], rank=0, world_size=1, process_group=None):
        """sample_and_analyze (pharmacodiff.py:320-357) on explicit pockets; with a process group the
        validity numerator/denominator are summed over ranks with one all-reduce (RCCL on GPUs)."""
        n_pharms = [[g.num_nodes('pharm')] * self.pharms_per_pocket for g in pockets]
        coms = torch.stack([g.pharm_x0.mean(dim=0) for g in pockets], dim=0)
        was_training = self.training
        self.eval()
        sampled = self.sample(pockets, n_pharms, max_batch_size=64, init_pharm_com=coms, rank=rank, world_size=world_size)
        self.train(was_training)
        flat = [p for pocket in sampled for p in pocket]
        return SampleAnalyzer().analyze(flat, process_group=process_group, device=self.device)


def model_from_config(config: dict, ckpt=None) -> PharmacophoreDiff:
    """pharmacoforge/config_utils/load_from_config.py:6-32 (same yaml -> constructor mapping)."""
    ev = config['training']['evaluation']
    return PharmacophoreDiff(
        pharm_nf=len(config['dataset']['ph_type_map']), rec_nf=len(config['dataset']['prot_elements']),
        ph_type_map=config['dataset']['ph_type_map'], processed_data_dir=config['dataset']['processed_data_dir'],
        n_pockets_to_sample=ev['n_pockets'], pharms_per_pocket=ev['pharms_per_pocket'],
        sample_interval=ev['sample_interval'], val_loss_interval=ev['val_loss_interval'],
        batch_size=config['training']['batch_size'], graph_config=config['graph'], dynamics_config=config['dynamics'],
        lr_scheduler_config=config['lr_scheduler'], **config['diffusion'])
