"""PfEngine: one libpfdyn handle per (process, GPU).  Marshals torch tensors (device memory owned
by PyTorch) and the current torch HIP stream into the C ABI of include/pfdyn.h."""
import ctypes
from typing import Dict, Optional

import torch

from . import _lib as L


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous CUDA/HIP tensors"
    return ctypes.c_void_p(t.data_ptr())


def _i32_host(t: torch.Tensor):
    """contiguous int32 numpy copy of an index tensor (host)."""
    import numpy as np
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class PfEngine:
    def __init__(self, *, pharm_nf=6, rec_nf=11, vector_size=16, n_hidden_scalars=128, n_convs=2,
                 n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4, message_norm="mean", ff_k=0, pf_k=5,
                 graph_cutoffs=None, rbf_dmax=15.0, rbf_dim=16, device=None):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.PfError("no HIP device visible to PyTorch: libpfdyn has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        cut = {"pp": 3.5, "pf": 8.0, "fp": 8.0, "ff": 9.0}
        cut.update(graph_cutoffs or {})
        if isinstance(message_norm, dict):
            raise L.PfError("dict-valued message_norm is unusable in the reference too (gvp.py:453)")
        if message_norm == "mean":
            mode, val = L.PF_NORM_MEAN, 1.0
        elif float(message_norm) == 0.0:
            mode, val = L.PF_NORM_GRAPH, 0.0
        else:
            mode, val = L.PF_NORM_VALUE, float(message_norm)
        self.cfg = L.PfConfig(L.PF_ABI_VERSION, pharm_nf, rec_nf, vector_size, n_hidden_scalars, n_convs,
                              n_message_gvps, n_update_gvps, n_noise_gvps, mode, val, int(ff_k), int(pf_k),
                              float(cut["pp"]), float(cut["pf"]), float(cut["fp"]), float(cut["ff"]),
                              float(rbf_dmax), int(rbf_dim))
        self.pharm_nf, self.rec_nf = pharm_nf, rec_nf
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.pf_create(ctypes.byref(self.cfg), ctypes.byref(self._h))
        if rc < 0:
            msg = self.lib.pf_last_error(None)
            raise L.PfError(f"pf_create failed ({rc}): {msg.decode() if msg else ''}")
        self.Nf = self.Np = self.B = 0
        self._keep = []

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self.lib.pf_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def _ck(self, rc, what):
        return L.check(self.lib, self._h, rc, what)

    # -- weights (reference state_dict key layout) -------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "dynamics."):
        """sd: reference keys (``dynamics.*``); ``prefix`` is what the given keys start with
        relative to the reference layout (pass prefix='' for a bare PharmRecDynamicsGVP dict)."""
        with torch.cuda.device(self.device):
            for k, v in sd.items():
                name = k if prefix == "dynamics." else "dynamics." + k
                if not name.startswith("dynamics."):
                    continue           # e.g. gamma.gamma
                t = v.detach().to("cpu", torch.float32).contiguous()
                shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
                self._ck(self.lib.pf_set_weight(self._h, name.encode(), ctypes.c_void_p(t.data_ptr()) if t.numel() else None,
                                                t.dim(), shape), f"pf_set_weight({name})")
            self._ck(self.lib.pf_commit_weights(self._h), "pf_commit_weights")

    # -- static batch --------------------------------------------------------------------------
    def _upload(self, t: torch.Tensor) -> torch.Tensor:
        """fp32 device copy of a host tensor through pinned memory (a pageable host -> device copy waits for everything
        already enqueued on the stream, i.e. for the previous batch); device tensors pass through."""
        if t.is_cuda:
            return _f32(t, self.device)
        return t.detach().to(torch.float32).contiguous().pin_memory().to(self.device, non_blocking=True)

    def set_batch(self, prot_x, prot_h, prot_ptr, pharm_ptr, pp_src, pp_dst, pocket_uid=None):
        """pf_set_pocket_batch / pf_set_pocket_batch_host: asynchronous on the current stream.  ``pocket_uid`` ([B]
        integers, optional): graphs with equal values are claimed to be copies of one pocket (PocketGraph.pocket_uid), which
        lets them share conv layer 0's protein-protein messages during sampling.  The claim is verified before it is used:
        the library checks atom counts, pp in-degrees and pp sources of every copy, and coordinates and features byte for
        byte when the pocket tensors are host-resident; for device-resident tensors (which the library never reads on the
        host) the coordinate / feature comparison is done here, on the device.  A false claim raises PfError."""
        import numpy as np
        # index arrays: int32 host copies through numpy (single-threaded; a torch dtype conversion of half a million
        # elements goes through the intra-op thread pool, whose wake-up stalls for tens of milliseconds now and then on
        # hosts with hundreds of hardware threads)
        pptr, fptr, src, dst = (_i32_host(t) for t in (prot_ptr, pharm_ptr, pp_src, pp_dst))
        on_host = not prot_x.is_cuda and not prot_h.is_cuda
        rep = None
        if pocket_uid is not None:
            uid = np.asarray(pocket_uid.detach().cpu().numpy() if isinstance(pocket_uid, torch.Tensor) else pocket_uid).reshape(-1)
            first = {}
            rep = np.asarray([first.setdefault(int(u), i) for i, u in enumerate(uid)], dtype=np.int32)
            if len(first) == rep.size:
                rep = None                                # no pocket has copies
        if rep is not None and not on_host and rep.size == pptr.size - 1:
            cnt = np.diff(pptr)
            if np.array_equal(cnt, cnt[rep]):             # (unequal atom counts: the library refuses the claim itself)
                # atom i of a copy <-> the same atom of its representative: one gather and one comparison on the device
                idx = np.concatenate([np.arange(pptr[r], pptr[r + 1]) for r in rep]) if rep.size else np.zeros(0, np.int64)
                idx_t = torch.from_numpy(idx.astype(np.int64)).to(prot_x.device)
                same = bool(torch.equal(prot_x, prot_x[idx_t])) and bool(torch.equal(prot_h.to(prot_x.device), prot_h.to(prot_x.device)[idx_t]))
                if not same:
                    raise L.PfError("pf_set_pocket_groups: a graph is not a copy of the pocket it names (coordinates / features differ)")
        # the claim always travels with its bind (an empty one clears whatever an earlier, failed bind left behind)
        if rep is not None:
            self._ck(self.lib.pf_set_pocket_groups(self._h, int(rep.size), rep.ctypes.data), "pf_set_pocket_groups")
        else:
            self._ck(self.lib.pf_set_pocket_groups(self._h, 0, None), "pf_set_pocket_groups")
        self.B = int(pptr.size - 1)
        self.Np, self.Nf = int(pptr[-1]), int(fptr[-1])
        with torch.cuda.device(self.device):
            args = (int(src.size), src.ctypes.data if src.size else None, dst.ctypes.data if dst.size else None, _stream_ptr())
            if on_host:
                # host-resident pockets (the sampling drivers): the library stages them with its tables -- one upload, the
                # one-hot check on the host copy, nothing waits for the device
                hx = np.ascontiguousarray(prot_x.detach().numpy(), dtype=np.float32)
                hh = np.ascontiguousarray(prot_h.detach().numpy(), dtype=np.float32)
                self._ck(self.lib.pf_set_pocket_batch_host(self._h, self.B, pptr.ctypes.data, fptr.ctypes.data, hx.ctypes.data,
                                                           hh.ctypes.data, *args), "pf_set_pocket_batch_host")
            else:
                px, ph = self._upload(prot_x), self._upload(prot_h)
                self._ck(self.lib.pf_set_pocket_batch(self._h, self.B, pptr.ctypes.data, fptr.ctypes.data, _dptr(px), _dptr(ph),
                                                      *args), "pf_set_pocket_batch")

    def build_pp_edges(self, prot_x, prot_ptr, max_num_neighbors=100):
        px = _f32(prot_x, self.device)
        pptr = prot_ptr.to("cpu", torch.int32).contiguous()
        B = int(pptr.numel() - 1)
        with torch.cuda.device(self.device):
            n = self._ck(self.lib.pf_build_pp_edges(self._h, B, pptr.data_ptr(), _dptr(px), max_num_neighbors,
                                                    None, None, 0, _stream_ptr()), "pf_build_pp_edges")
            src = torch.zeros(max(n, 1), dtype=torch.int32)
            dst = torch.zeros(max(n, 1), dtype=torch.int32)
            self._ck(self.lib.pf_build_pp_edges(self._h, B, pptr.data_ptr(), _dptr(px), max_num_neighbors,
                                                src.data_ptr(), dst.data_ptr(), n, _stream_ptr()), "pf_build_pp_edges")
        return src[:n].long(), dst[:n].long()

    # -- the boundary function -----------------------------------------------------------------
    def dynamics(self, pharm_x, pharm_h, t, prot_x=None):
        x, hh, tt = _f32(pharm_x, self.device), _f32(pharm_h, self.device), _f32(t, self.device)
        px = _f32(prot_x, self.device) if prot_x is not None else None
        eps_h = torch.empty(self.Nf, self.pharm_nf, device=self.device)
        eps_x = torch.empty(self.Nf, 3, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_dynamics_forward(self._h, _dptr(px), _dptr(x), _dptr(hh), _dptr(tt), _dptr(eps_h),
                                                  _dptr(eps_x), _stream_ptr()), "pf_dynamics_forward")
        return eps_h, eps_x

    # -- training step (gradients of the boundary function) -------------------------------------
    def param_layout(self):
        """[(reference key, offset, numel)] of the flat parameter / gradient vector (state-dict order)."""
        n = ctypes.c_int64()
        nt = ctypes.c_int32()
        self._ck(self.lib.pf_param_count(self._h, ctypes.byref(n), ctypes.byref(nt)), "pf_param_count")
        out = []
        for i in range(nt.value):
            name = ctypes.c_char_p()
            off = ctypes.c_int64()
            cnt = ctypes.c_int64()
            self._ck(self.lib.pf_param_layout(self._h, i, ctypes.byref(name), ctypes.byref(off), ctypes.byref(cnt)), "pf_param_layout")
            out.append((name.value.decode(), off.value, cnt.value))
        self.n_params = n.value
        return out

    def train_forward(self, pharm_x, pharm_h, t, prot_x=None, dropout=0.0, seed=0):
        """PharmRecDynamicsGVP.forward in train() mode; keeps the per-layer state for train_backward."""
        x, hh, tt = _f32(pharm_x, self.device), _f32(pharm_h, self.device), _f32(t, self.device)
        px = _f32(prot_x, self.device) if prot_x is not None else None
        eps_h = torch.empty(self.Nf, self.pharm_nf, device=self.device)
        eps_x = torch.empty(self.Nf, 3, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_train_forward(self._h, _dptr(px), _dptr(x), _dptr(hh), _dptr(tt), float(dropout),
                                               int(seed) & 0xFFFFFFFF, _dptr(eps_h), _dptr(eps_x), _stream_ptr()),
                     "pf_train_forward")
        return eps_h, eps_x

    def train_backward(self, g_eps_h, g_eps_x):
        """d(loss)/d(parameters) as one flat vector (see param_layout) given d(loss)/d(eps)."""
        gh, gx = _f32(g_eps_h, self.device), _f32(g_eps_x, self.device)
        if not hasattr(self, "n_params"):
            self.param_layout()
        grad = torch.empty(self.n_params, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_train_backward(self._h, _dptr(gh), _dptr(gx), _dptr(grad), _stream_ptr()), "pf_train_backward")
        return grad

    def train_loss_forward(self, pharm_x0, pharm_h0, t_int, eps_x, eps_h, alpha_tab, sigma_tab, n_timesteps, feat_norm,
                           remove_com=True, weighted_loss=False, dropout=0.0, seed=0):
        """PharmacophoreDiff.forward (pharmacodiff.py:162-243, noise parameterisation) around the bound batch as one call:
        COM removal, noising, the train-mode dynamics, losses and metrics.  Returns a device tensor [9]: pos loss, feat
        loss, position error, weighted position error, accuracy, weighted accuracy, then what a step derives from them --
        total loss, total error, weighted total error."""
        x0, h0 = _f32(pharm_x0, self.device), _f32(pharm_h0, self.device)
        ex, eh = _f32(eps_x, self.device), _f32(eps_h, self.device)
        ti = t_int.to(self.device, torch.int32).contiguous()
        al, sg = _f32(alpha_tab, self.device), _f32(sigma_tab, self.device)
        out = torch.empty(9, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_train_loss_forward(self._h, _dptr(x0), _dptr(h0), _dptr(ti), _dptr(ex), _dptr(eh), _dptr(al),
                                                    _dptr(sg), int(n_timesteps), float(feat_norm), int(bool(remove_com)),
                                                    int(bool(weighted_loss)), float(dropout), int(seed) & 0xFFFFFFFF,
                                                    _dptr(out), _stream_ptr()), "pf_train_loss_forward")
        return out

    def _grad_out(self, out):
        if not hasattr(self, "n_params"):
            self.param_layout()
        if out is None:
            return torch.empty(self.n_params, device=self.device)
        if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != self.n_params or out.device != self.device:
            raise ValueError("`out` must be a contiguous fp32 vector of n_params elements on the engine's device")
        return out

    def train_loss_backward(self, g_pos, g_feat, out=None):
        """d(g_pos * pos loss + g_feat * feat loss)/d(parameters) of the last train_loss_forward, as one flat vector.
        out: an existing flat fp32 device vector to receive the gradient (every element is stored, nothing is accumulated)."""
        gp, gf = _f32(g_pos, self.device).reshape(1), _f32(g_feat, self.device).reshape(1)
        grad = self._grad_out(out)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_train_loss_backward(self._h, _dptr(gp), _dptr(gf), _dptr(grad), _stream_ptr()),
                     "pf_train_loss_backward")
        return grad

    def train_loss_backward_out(self, g_out, out=None):
        """The same from the upstream gradient of train_loss_forward's whole output vector ([9]; entries 0, 1 and 6 count).
        out: an existing flat fp32 device vector to receive the gradient (every element is stored, nothing is accumulated)."""
        go = _f32(g_out, self.device).reshape(9)
        grad = self._grad_out(out)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_train_loss_backward_out(self._h, _dptr(go), _dptr(grad), _stream_ptr()),
                     "pf_train_loss_backward_out")
        return grad

    def set_flat_params(self, flat):
        """Replace every parameter from one flat device vector (param_layout order); the packed kernel weights are
        refreshed by a device-side gather."""
        f = _f32(flat, self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_set_flat_params(self._h, _dptr(f), _stream_ptr()), "pf_set_flat_params")

    def get_flat_params(self):
        if not hasattr(self, "n_params"):
            self.param_layout()
        out = torch.empty(self.n_params, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_get_flat_params(self._h, _dptr(out), _stream_ptr()), "pf_get_flat_params")
        return out

    def adam_step(self, params, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """Fused Adam on flat device vectors (in place) + refresh of the engine's weights from ``params``."""
        for t in (params, grad, exp_avg, exp_avg_sq):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == self.n_params
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_adam_step(self._h, _dptr(params), _dptr(grad), _dptr(exp_avg), _dptr(exp_avg_sq), int(step),
                                           float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                           _stream_ptr()), "pf_adam_step")

    def set_train_precision(self, precision):
        """'f32' (default; what the reference trains in) or 'bf16' (the labelled bf16 leg: dense Linears of the message chains'
        forward and of the gradient kernels on bf16 matrix instructions, fp32 accumulation, fp32 master weights)."""
        code = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}.get(str(precision).lower())
        if code is None:
            raise ValueError(f"train precision must be 'f32' or 'bf16', got {precision!r}")
        self._ck(self.lib.pf_train_set_precision(self._h, code), "pf_train_set_precision")

    def train_precision(self):
        out = ctypes.c_int32(0)
        self._ck(self.lib.pf_train_get_precision(self._h, ctypes.byref(out)), "pf_train_get_precision")
        return "bf16" if out.value == 1 else "f32"

    def set_dropout_masks(self, masks):
        """tests: [n_convs, 2, N, 144] multipliers used instead of the built-in generator (None restores it)."""
        self._mask_keepalive = None if masks is None else _f32(masks, self.device)
        self._ck(self.lib.pf_debug_set_dropout_masks(self._h, _dptr(self._mask_keepalive)), "pf_debug_set_dropout_masks")

    def dropout_mask(self, layer, which, dropout, seed):
        """[N, 144] multipliers of conv layer ``layer`` (which: 0 message, 1 residual dropout); rows are global node
        ids (protein atoms first), columns 128 scalar features then 16 vector channels."""
        out = torch.empty(self.Np + self.Nf, 144, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_debug_dropout_mask(self._h, layer, which, float(dropout), int(seed) & 0xFFFFFFFF,
                                                    _dptr(out), _stream_ptr()), "pf_debug_dropout_mask")
        return out

    # -- sampling ------------------------------------------------------------------------------
    @staticmethod
    def coef_array(coef: Dict[str, torch.Tensor], order):
        """coef: per-s tensors (host logic of the diffusion wrapper); order: iterable of s."""
        order = list(order)
        arr = (L.PfStepCoef * max(len(order), 1))()
        for i, s in enumerate(order):
            arr[i] = L.PfStepCoef(float(coef["t"][s]), float(coef["alpha_t_given_s"][s]), float(coef["var_terms"][s]),
                                  float(coef["sigma"][s]), float(coef["ep_zt"][s]), float(coef["ep_pred"][s]))
        return arr

    def sample_begin(self, noise0, init_pharm_com=None):
        nz = _f32(noise0, self.device)
        com = _f32(init_pharm_com, self.device) if init_pharm_com is not None else None
        self._keep = [nz, com]
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_sample_begin(self._h, _dptr(com), _dptr(nz), _stream_ptr()), "pf_sample_begin")

    def prepare_timesteps(self, coef_arr, n=None):
        """Optional before a loop of denoise_step calls: the timesteps the loop will visit (pf_prepare_timesteps)."""
        n = len(coef_arr) if n is None else n
        tv = (ctypes.c_float * max(n, 1))(*[coef_arr[i].t for i in range(n)])
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_prepare_timesteps(self._h, tv, n, _stream_ptr()), "pf_prepare_timesteps")

    def denoise_step(self, coef_struct, noise, ep_coord=False, ep_feat=False):
        nz = _f32(noise, self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_denoise_step(self._h, ctypes.byref(coef_struct), _dptr(nz), int(ep_coord), int(ep_feat),
                                              _stream_ptr()), "pf_denoise_step")

    def sample_frame(self, feat_norm_constant=1.0):
        x = torch.empty(self.Nf, 3, device=self.device)
        hh = torch.empty(self.Nf, self.pharm_nf, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_sample_frame(self._h, float(feat_norm_constant), _dptr(x), _dptr(hh), _stream_ptr()),
                     "pf_sample_frame")
        return x, hh

    def sample(self, coef_arr, n_steps, noise, init_pharm_com=None, ep_coord=False, ep_feat=False,
               feat_norm_constant=1.0, trajectory=False):
        nz = _f32(noise, self.device)
        assert nz.shape[0] >= n_steps + 1 and nz.shape[1] == self.Nf and nz.shape[2] == 3 + self.pharm_nf
        com = _f32(init_pharm_com, self.device) if init_pharm_com is not None else None
        x0 = torch.empty(self.Nf, 3, device=self.device)
        h0 = torch.empty(self.Nf, self.pharm_nf, device=self.device)
        tx = th = None
        if trajectory:
            tx = torch.empty(n_steps + 1, self.Nf, 3, device=self.device)
            th = torch.empty(n_steps + 1, self.Nf, self.pharm_nf, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_sample(self._h, n_steps, coef_arr, _dptr(nz), _dptr(com), int(ep_coord), int(ep_feat),
                                        float(feat_norm_constant), _dptr(x0), _dptr(h0), _dptr(tx), _dptr(th),
                                        _stream_ptr()), "pf_sample")
        return (x0, h0, tx, th) if trajectory else (x0, h0)

    def sample_status(self):
        """Validity of the sampling runs that ended on this handle since the last call (pf_sample_status): raises PfError when a
        hand-over inside a merged launch timed out.  Call once the results have been waited for (stream / event synchronisation)
        and before using them; touches no device."""
        n = ctypes.c_int32()
        self._ck(self.lib.pf_sample_status(self._h, ctypes.byref(n)), "pf_sample_status")

    # -- introspection -------------------------------------------------------------------------
    def get_edges(self, etype: int):
        with torch.cuda.device(self.device):
            n = self._ck(self.lib.pf_debug_get_edges(self._h, etype, None, None, 0, _stream_ptr()), "pf_debug_get_edges")
            src = torch.zeros(max(n, 1), dtype=torch.int32)
            dst = torch.zeros(max(n, 1), dtype=torch.int32)
            self._ck(self.lib.pf_debug_get_edges(self._h, etype, src.data_ptr(), dst.data_ptr(), n, _stream_ptr()),
                     "pf_debug_get_edges")
        return src[:n].long(), dst[:n].long()

    def conv_layer(self, layer, prot_x, pharm_x, h_prot, v_prot, h_pharm, v_pharm):
        a = [_f32(t, self.device) for t in (prot_x, pharm_x, h_prot, v_prot, h_pharm, v_pharm)]
        outs = [torch.empty_like(a[2]), torch.empty_like(a[3]), torch.empty_like(a[4]), torch.empty_like(a[5])]
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_debug_conv_layer(self._h, layer, *[_dptr(t) for t in a], *[_dptr(t) for t in outs],
                                                  _stream_ptr()), "pf_debug_conv_layer")
        return outs

    def work(self):
        """(reference-equivalent flops, bytes, [ff, pf, fp, pp] edge counts) of the last dynamics call."""
        w = self.work_detail()
        return w["flops"], w["bytes"], w["edges"]

    def work_detail(self):
        fl, by, ex = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        ne = (ctypes.c_int64 * 4)()
        el = (ctypes.c_int64 * max(int(self.cfg.n_convs), 1))()
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_debug_work(self._h, ctypes.byref(fl), ctypes.byref(by), ne, ctypes.byref(ex), el,
                                            _stream_ptr()), "pf_debug_work")
        return {"flops": fl.value, "bytes": by.value, "edges": list(ne), "executed_flops": ex.value,
                "executed_edges_per_layer": list(el)}

    def counts(self):
        """Row / edge counts of the last dynamics call: dict(ff, pf, fp, pp, pa, active_atoms, centers, atoms)."""
        out = (ctypes.c_int64 * 8)()
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_debug_counts(self._h, out, _stream_ptr()), "pf_debug_counts")
        return dict(zip(("ff", "pf", "fp", "pp", "pa", "active_atoms", "centers", "atoms"), list(out)))

    def kernel_family(self, layer: int = 0) -> int:
        """Rows per wave of the edge-message launch of `layer` in the last dynamics call: 4 / 8 = row-group kernels
        (k_rg_edge), 16 = 16-row items on four waves (k_n16_edge; 17: the fused launch k_n16_fused, which also computes conv layer 0's
        node update of its source rows), 32 = one wave per tile (k_edge_msg), 128 = four waves per tile (k_edge_msg_coop)."""
        r = ctypes.c_int32()
        self._ck(self.lib.pf_debug_kernel_family(self._h, int(layer), ctypes.byref(r)), "pf_debug_kernel_family")
        return int(r.value)

    def last_eps(self):
        """(eps_h, eps_x) of the last denoising step's dynamics call."""
        eh = torch.empty(self.Nf, self.pharm_nf, device=self.device); ex = torch.empty(self.Nf, 3, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_debug_last_eps(self._h, _dptr(eh), _dptr(ex), _stream_ptr()), "pf_debug_last_eps")
        return eh, ex

    def xchg_timeouts(self) -> int:
        """Time-outs of the merged last launch's exchange (k_rg_node_hs_build) since the handle was created; synchronises the device.
        The product-path check is sample_status()."""
        r = ctypes.c_int32()
        self._ck(self.lib.pf_debug_xchg_timeouts(self._h, ctypes.byref(r)), "pf_debug_xchg_timeouts")
        return int(r.value)

    def xchg_fault(self, drop_word: bool, poll_max: int = 0):
        """Diagnostic (pf_debug_xchg_fault): make the merged launch's producers skip one exchange word / shorten the poll bound."""
        self._ck(self.lib.pf_debug_xchg_fault(self._h, int(bool(drop_word)), int(poll_max)), "pf_debug_xchg_fault")

    def ahead(self):
        """What the last denoising step's merged launch did ahead for the next call (pf_debug_ahead): dict(pa_skipped, pa_ahead,
        center_hoist, centers)."""
        out = (ctypes.c_int64 * 4)()
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_debug_ahead(self._h, out, _stream_ptr()), "pf_debug_ahead")
        return dict(zip(("pa_skipped", "pa_ahead", "center_hoist", "centers"), list(out)))

    def l0_hoist(self) -> int:
        """Rows per hoisted wave of conv layer 0's pp messages in the last dynamics call (0: static hoist not used)."""
        r = ctypes.c_int32()
        self._ck(self.lib.pf_debug_l0_hoist(self._h, ctypes.byref(r)), "pf_debug_l0_hoist")
        return int(r.value)

    def debug_chain(self, kind: int, layer: int, sub: int, s_in: torch.Tensor, v_in: torch.Tensor):
        """pf_debug_chain: one chain of the row-group kernels on the given rows (kinds and shapes: include/pfdyn.h)."""
        s_in = s_in.to(self.device, torch.float32).contiguous()
        v_in = v_in.to(self.device, torch.float32).contiguous()
        n = s_in.shape[0]
        if kind == 3:
            s_out = torch.empty(n, self.cfg.pharm_nf, device=self.device)
            v_out = torch.empty(n, 3, device=self.device)
        else:
            s_out = torch.empty(n, 128, device=self.device)
            v_out = torch.empty(n, 16, 3, device=self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_debug_chain(self._h, int(kind), int(layer), int(sub), int(n), _dptr(s_in), _dptr(v_in),
                                             _dptr(s_out), _dptr(v_out), _stream_ptr()), "pf_debug_chain")
        return s_out, v_out

    KERNEL_CLASSES = ("encode", "build_edges", "edge_msg", "node_update", "noise_head", "step_update", "edge_msg_coop",
                      "node_update_coop", "edge_msg_last")

    def profile_enable(self, mask: int):
        self._ck(self.lib.pf_profile_enable(self._h, mask), "pf_profile_enable")

    TRAIN_KERNEL_CLASSES = ("bwd_head", "bwd_node", "bwd_edge_level", "bwd_rest")

    def profile_read_train(self):
        """{gradient-kernel class: (total device ms, launches)} since the last read (profile_enable bits 9..12)."""
        ms = (ctypes.c_double * 4)()
        n = (ctypes.c_int64 * 4)()
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_profile_read_train(self._h, ms, n, _stream_ptr()), "pf_profile_read_train")
        return {k: (ms[i], n[i]) for i, k in enumerate(self.TRAIN_KERNEL_CLASSES)}

    def profile_read(self):
        """{kernel class: (total device ms, launches)} since the last enable/read (synchronises)."""
        ms = (ctypes.c_double * 9)()
        n = (ctypes.c_int64 * 9)()
        with torch.cuda.device(self.device):
            self._ck(self.lib.pf_profile_read(self._h, ms, n, _stream_ptr()), "pf_profile_read")
        return {k: (ms[i], n[i]) for i, k in enumerate(self.KERNEL_CLASSES)}
