"""Gradient parity of the HIP training path (pf_train_forward / pf_train_backward) against the oracle's autograd
and the reference's own gradients (tests/golden/train_grads*.npz)."""
import pytest
import torch

from oracle import pf_oracle as O
from helpers import GRAD_CASES, batch_from, load

pytestmark = pytest.mark.gpu


def make_engine(cfg, sd, batch):
    import pharmacoforge_amd as pfa
    eng = pfa.PfEngine(pharm_nf=cfg.pharm_nf, rec_nf=cfg.rec_nf, n_convs=cfg.n_convs,
                       n_message_gvps=cfg.n_message_gvps, n_update_gvps=cfg.n_update_gvps,
                       n_noise_gvps=cfg.n_noise_gvps, message_norm=cfg.message_norm, ff_k=cfg.ff_k, pf_k=cfg.pf_k,
                       graph_cutoffs={"pp": cfg.cutoff_pp, "pf": cfg.cutoff_pf, "fp": cfg.cutoff_fp, "ff": cfg.cutoff_ff})
    eng.load_state_dict(sd)
    eng.set_batch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst)
    return eng


def noised_inputs(cfg, batch, z, T):
    """x_t, h_t, prot_x, t of PharmacophoreDiff.forward (pharmacodiff.py:179-199), via the oracle's algebra."""
    bidx = batch.batch_idxs()
    com = O.segment_mean(z["x0"], batch.pharm_ptr)
    x0 = z["x0"] - com[bidx["pharm"]]
    prot_x = batch.prot_x - com[bidx["prot"]]
    t = z["t_int"].float() / T
    gamma = O.gamma_table(T, 1e-5)
    gt = O.gamma_lookup(gamma, t, T)
    a = O.alpha(gt)[bidx["pharm"]][:, None]
    s = O.sigma(gt)[bidx["pharm"]][:, None]
    x_t = a * x0 + s * z["eps_x"]
    h_t = a * z["h0"] + s * z["eps_h"]
    c = O.segment_mean(x_t, batch.pharm_ptr)
    return x_t - c[bidx["pharm"]], h_t, prot_x - c[bidx["prot"]], t


def masks_from_engine(eng, cfg, p, seed, Np, Nf):
    out = []
    for layer in range(cfg.n_convs):
        m0 = eng.dropout_mask(layer, 0, p, seed).cpu()
        m1 = eng.dropout_mask(layer, 1, p, seed).cpu()
        d = {}
        for nt, sl in (("prot", slice(0, Np)), ("pharm", slice(Np, Np + Nf))):
            d[nt] = (m0[sl, :128], m0[sl, 128:], m1[sl, :128], m1[sl, 128:])
        out.append(d)
    return out


def flat_to_dict(eng, grad):
    g = grad.cpu()
    return {name: g[off:off + n] for name, off, n in eng.param_layout()}


def compare(got, ref, tol, what):
    bad = []
    for k, r in ref.items():
        if r.numel() == 0:
            continue
        g = got[k].reshape(r.shape)
        scale = float(r.abs().max())
        err = float((g - r).abs().max())
        if err > tol * scale + 1e-7:
            bad.append((k, err, scale))
    assert not bad, (what, bad[:8], len(bad))


@pytest.mark.parametrize("name", sorted(GRAD_CASES))
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_gradients_vs_oracle(name, p_drop):
    z = load(name)
    cfg = GRAD_CASES[name]
    batch = batch_from(z)
    T = int(z["T"])
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    eng = make_engine(cfg, sd, batch)
    x_t, h_t, prot_x, t = noised_inputs(cfg, batch, z, T)
    Np, Nf = int(batch.prot_ptr[-1]), int(batch.pharm_ptr[-1])
    seed = 1234
    eps_h, eps_x = eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=p_drop, seed=seed)
    drop = masks_from_engine(eng, cfg, p_drop, seed, Np, Nf) if p_drop > 0 else None
    if drop is not None:
        keep = torch.cat([m.reshape(-1) for d in drop for nt in d for m in d[nt]])
        assert set(keep.unique().tolist()) <= {0.0, float(torch.tensor(1.0 / (1.0 - p_drop), dtype=torch.float32))}
        frac = float((keep == 0).float().mean())
        assert abs(frac - p_drop) < 0.02, frac
    # oracle: same inputs, same masks; loss = sum of the two MSE-style terms of pharmacodiff.py:208-232
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    with torch.enable_grad():
        oh, ox = O.dynamics_forward(leaf, cfg, batch, prot_x, x_t, h_t, t, dropout=drop)
        loss = (z["eps_x"] - ox).square().sum() / z["eps_x"].numel() + (z["eps_h"] - oh).square().sum() / z["eps_h"].numel()
        loss.backward()
    assert float((eps_h.cpu() - oh.detach()).abs().max()) < 2e-4
    assert float((eps_x.cpu() - ox.detach()).abs().max()) < 2e-4
    g_h = (-2.0 / z["eps_h"].numel()) * (z["eps_h"] - eps_h.cpu())
    g_x = (-2.0 / z["eps_x"].numel()) * (z["eps_x"] - eps_x.cpu())
    grad = eng.train_backward(g_h, g_x)
    got = flat_to_dict(eng, grad)
    ref = {k: (torch.zeros_like(v) if v.grad is None else v.grad) for k, v in leaf.items()}
    compare(got, ref, 2e-3, name)


EXTRA_CASES = {
    # per-graph message normalisation (gvp.py:504-507), kNN ff edges
    "per_graph_norm_knn": (O.DynamicsConfig(ff_k=2, pf_k=3, message_norm=0), [31, 32, 33], 40, [5, 4, 6]),
    # larger pockets: destination segments that span 32-slot tiles, several node tiles, radius pf edges
    "large_radius": (O.DynamicsConfig(pf_k=0, message_norm=10), [41, 42], 150, [8, 7]),
    # single GVP per chain, three conv layers
    "shallow_chains": (O.DynamicsConfig(n_convs=3, n_message_gvps=1, n_update_gvps=1, n_noise_gvps=2), [51], 64, [6]),
    # one conv layer (it is first and last at once); graphs with a single center (no ff edges at all)
    "single_layer_single_center": (O.DynamicsConfig(n_convs=1), [61, 62, 63], 33, [1, 1, 2]),
    # four conv layers (class default depth): two dense layers below the pruned one, sum aggregation
    "deep": (O.DynamicsConfig(n_convs=4, n_noise_gvps=3, message_norm=1, pf_k=0), [71, 72], 48, [4, 5]),
}


@pytest.mark.parametrize("name", sorted(EXTRA_CASES) + ["soft_features"])
def test_gradients_vs_oracle_more_configs(name):
    soft = name == "soft_features"
    cfg, seeds, n_prot, n_pharm = EXTRA_CASES["large_radius" if soft else name]
    batch = O.synthetic_batch(seeds, n_prot, n_pharm, cfg)
    if soft:
        # protein feature rows that are NOT element one-hots: the encoder backward then differentiates every atom on its own
        # (with one-hots it works on per-(graph, element) sums of the upstream gradient, k_enc_group)
        g0 = torch.Generator().manual_seed(5)
        batch = O.PocketBatch(batch.prot_x, batch.prot_h + 0.25 * torch.rand(batch.prot_h.shape, generator=g0), batch.prot_ptr,
                              batch.pharm_ptr, batch.pp_src, batch.pp_dst)
    sd = O.make_state_dict(cfg, 3)
    eng = make_engine(cfg, sd, batch)
    Np, Nf, B = int(batch.prot_ptr[-1]), int(batch.pharm_ptr[-1]), batch.batch_size
    gen = torch.Generator().manual_seed(11)
    bidx = batch.batch_idxs()
    com = O.segment_mean(batch.prot_x, batch.prot_ptr)
    prot_x = batch.prot_x - com[bidx["prot"]]
    x_t = 2.5 * torch.randn(Nf, 3, generator=gen)
    h_t = torch.randn(Nf, cfg.pharm_nf, generator=gen)
    t = torch.rand(B, generator=gen)
    w_h, w_x = torch.randn(Nf, cfg.pharm_nf, generator=gen), torch.randn(Nf, 3, generator=gen)
    p_drop, seed = 0.2, 99
    eps_h, eps_x = eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=p_drop, seed=seed)
    drop = masks_from_engine(eng, cfg, p_drop, seed, Np, Nf)
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    with torch.enable_grad():
        oh, ox = O.dynamics_forward(leaf, cfg, batch, prot_x, x_t, h_t, t, dropout=drop)
        ((oh * w_h).sum() + (ox * w_x).sum()).backward()
    assert float((eps_h.cpu() - oh.detach()).abs().max()) < 5e-4 * max(1.0, float(oh.detach().abs().max()))
    assert float((eps_x.cpu() - ox.detach()).abs().max()) < 5e-4 * max(1.0, float(ox.detach().abs().max()))
    got = flat_to_dict(eng, eng.train_backward(w_h, w_x))
    ref = {k: (torch.zeros_like(v) if v.grad is None else v.grad) for k, v in leaf.items()}
    compare(got, ref, 3e-3, name)
    # a second backward of the same forward reproduces the gradient BIT FOR BIT: per-block gradient copies summed in
    # block order, and the level-0 scatter to the source nodes on fixed-point accumulators (order-independent)
    again = flat_to_dict(eng, eng.train_backward(w_h, w_x))
    for k in got:
        assert torch.equal(again[k], got[k]), k
    # ... and so does a whole fresh forward + backward
    eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=p_drop, seed=seed)
    third = flat_to_dict(eng, eng.train_backward(w_h, w_x))
    for k in got:
        assert torch.equal(third[k], got[k]), k


@pytest.mark.parametrize("name", sorted(GRAD_CASES))
def test_pruned_and_dense_training_agree(name, monkeypatch):
    """Walking only the rows whose gradient can be non-zero (last layer: pharm side; layer before: active atoms) is
    exact: with PFDYN_NO_PRUNE every tile of every layer is walked and the gradients are the same up to rounding."""
    z = load(name)
    cfg = GRAD_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    x_t, h_t, prot_x, t = noised_inputs(cfg, batch, z, int(z["T"]))
    gen = torch.Generator().manual_seed(5)
    w_h, w_x = torch.randn(h_t.shape, generator=gen), torch.randn(x_t.shape, generator=gen)
    res = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("PFDYN_NO_PRUNE", "1")
            monkeypatch.setenv("PFDYN_NO_PRE", "1")
        eng = make_engine(cfg, sd, batch)
        eh, ex = eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=0.1, seed=77)
        res.append((eh.cpu(), ex.cpu(), flat_to_dict(eng, eng.train_backward(w_h, w_x))))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=2e-5, atol=2e-5)
    compare(res[0][2], {k: v for k, v in res[1][2].items()}, 1e-3, name)


@pytest.mark.parametrize("name", sorted(GRAD_CASES))
def test_training_forward_kernel_families_agree(name, monkeypatch):
    """The training forward runs on the row-group kernels by default (k_rg_edge<., SAVE> writing the per-level rows of the
    backward pass, k_rg_node with the two GVPDropout sites); PFDYN_TRAIN_TILE_EDGE=1 keeps the 32-slot tile edge kernel
    (k_edge_msg<., SAVE>) under the row-group node kernel, PFDYN_TRAIN_TILE_NODE=1 (or a batch beyond the row-group policy)
    the tile kernels for both; PFDYN_TRAIN_TILE_HEAD=1 runs the noise head on the tile kernel and lets k_bwd_head recompute its
    chain instead of reading the levels k_rg_unit<SAVE> left, PFDYN_TRAIN_NODE_RECOMPUTE=1 does the same for the update chains
    of k_bwd_node (k_rg_node<., SAVE>): same dropout masks (one hash), outputs and gradients equal up to
    summation order."""
    z = load(name)
    cfg = GRAD_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    x_t, h_t, prot_x, t = noised_inputs(cfg, batch, z, int(z["T"]))
    gen = torch.Generator().manual_seed(6)
    w_h, w_x = torch.randn(h_t.shape, generator=gen), torch.randn(x_t.shape, generator=gen)
    res = []
    for var in (None, "PFDYN_TRAIN_TILE_HEAD", "PFDYN_TRAIN_NODE_RECOMPUTE", "PFDYN_TRAIN_TILE_EDGE", "PFDYN_TRAIN_TILE_NODE"):
        if var:
            monkeypatch.setenv(var, "1")
        eng = make_engine(cfg, sd, batch)
        eh, ex = eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=0.15, seed=321)
        res.append((eh.cpu(), ex.cpu(), flat_to_dict(eng, eng.train_backward(w_h, w_x))))
        if var:
            monkeypatch.delenv(var)
    for other in res[1:]:
        torch.testing.assert_close(res[0][0], other[0], rtol=2e-4, atol=2e-4)
        torch.testing.assert_close(res[0][1], other[1], rtol=2e-4, atol=2e-4)
        compare(res[0][2], {k: v for k, v in other[2].items()}, 2e-3, name)


@pytest.mark.parametrize("name", sorted(GRAD_CASES))
def test_fused_fixed_point_join_equals_the_two_launch_form(name, monkeypatch):
    """Conv layer 0's fixed-point sums join G_h in the pass that also groups the protein rows by (graph, element) for the encoders'
    backward (k_fix_enc_group); PFDYN_NO_FIX_FUSE=1 runs k_fix_apply and k_enc_group one after the other.  Same per-element
    arithmetic, same summation order: every gradient is bitwise the same, and a second backward (which finds the accumulators
    cleared by the first) repeats it."""
    z = load(name)
    cfg = GRAD_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    x_t, h_t, prot_x, t = noised_inputs(cfg, batch, z, int(z["T"]))
    gen = torch.Generator().manual_seed(8)
    w_h, w_x = torch.randn(h_t.shape, generator=gen), torch.randn(x_t.shape, generator=gen)
    res = []
    for two in (False, True):
        if two:
            monkeypatch.setenv("PFDYN_NO_FIX_FUSE", "1")
        eng = make_engine(cfg, sd, batch)
        eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=0.1, seed=99)
        first = flat_to_dict(eng, eng.train_backward(w_h, w_x))
        again = flat_to_dict(eng, eng.train_backward(w_h, w_x))
        for k in first:
            assert torch.equal(first[k], again[k]), k
        res.append(first)
    assert any(k.endswith("prot_encoder.0.weight") and float(v.abs().max()) > 0 for k, v in res[0].items())
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize("name", sorted(GRAD_CASES))
def test_fixed_shape_level_kernels_equal_the_table_driven_form(name, monkeypatch):
    """k_bwd_edge_level is instantiated with the two message-GVP shapes of the architecture as compile-time constants
    (BwdEdgeLevelParams::fx); PFDYN_NO_FIXED_SHAPES=1 launches the form that reads the shape from the GVP table.  Same products in
    the same order: the gradients agree to rounding of differently contracted multiply-adds."""
    z = load(name)
    cfg = GRAD_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    x_t, h_t, prot_x, t = noised_inputs(cfg, batch, z, int(z["T"]))
    gen = torch.Generator().manual_seed(9)
    w_h, w_x = torch.randn(h_t.shape, generator=gen), torch.randn(x_t.shape, generator=gen)
    res = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("PFDYN_NO_FIXED_SHAPES", "1")
        eng = make_engine(cfg, sd, batch)
        eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=0.1, seed=123)
        res.append(flat_to_dict(eng, eng.train_backward(w_h, w_x)))
    compare(res[0], res[1], 2e-5, name)


def _cos_rel(got, ref):
    a, b = ref.double().reshape(-1), got.double().reshape(-1)
    na = float(a.norm())
    return float((a * b).sum() / (na * float(b.norm()) + 1e-300)), float((a - b).norm() / (na + 1e-300))


@pytest.mark.parametrize("name", sorted(GRAD_CASES))
def test_bf16_leg_gradients_against_the_fp32_path(name):
    """The bf16 training leg (pf_train_set_precision(PF_TRAIN_BF16): to_feats_out / gate products of the message chains' forward
    and of the gradient kernels on bf16 matrix instructions, fp32 accumulation) has no reference counterpart -- the reference
    trains in fp32 (pharmacodiff.py:162-263).  Contract on the golden cases, same handle, same dropout masks: against the fp32
    path (itself held to the oracle and the reference's gradients above) every tensor's gradient has cosine >= 0.999 and relative
    L2 error <= 2e-2, the whole gradient vector cosine >= 0.9999; switching back to fp32 reproduces the fp32 gradients bit for
    bit.  (tests/test_gpu_api.py holds the bf16 leg to the reference's own gradients, tests/test_gpu_fullsize.py to the oracle at
    batch 256.)"""
    z = load(name)
    cfg = GRAD_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    x_t, h_t, prot_x, t = noised_inputs(cfg, batch, z, int(z["T"]))
    eng = make_engine(cfg, sd, batch)
    gen = torch.Generator().manual_seed(6)
    w_h, w_x = torch.randn(h_t.shape, generator=gen), torch.randn(x_t.shape, generator=gen)
    res = {}
    for mode in ("f32", "bf16", "f32 again"):
        eng.set_train_precision(mode.split()[0])
        assert eng.train_precision() == mode.split()[0]
        eh, ex = eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=0.1, seed=77)
        res[mode] = (eh.cpu(), ex.cpu(), eng.train_backward(w_h, w_x).cpu())
    assert torch.equal(res["f32"][2], res["f32 again"][2])
    assert not torch.equal(res["f32"][2], res["bf16"][2])            # the bf16 kernels really ran
    for q in (0, 1):
        assert float((res["bf16"][q] - res["f32"][q]).norm() / res["f32"][q].norm()) <= 5e-3
    g32, g16 = res["f32"][2], res["bf16"][2]
    live = 0
    for k, off, n in eng.param_layout():
        r = g32[off:off + n]
        if n == 0 or float(r.abs().max()) == 0.0:
            continue
        c, e = _cos_rel(g16[off:off + n], r)
        live += 1
        assert c >= 0.999 and e <= 2e-2, (name, k, c, e)
    assert live >= 150
    assert _cos_rel(g16, g32)[0] >= 0.9999


def test_bf16_leg_rejects_a_backward_of_the_other_precision():
    z = load(sorted(GRAD_CASES)[0])
    cfg = GRAD_CASES[sorted(GRAD_CASES)[0]]
    batch = batch_from(z)
    eng = make_engine(cfg, O.make_state_dict(cfg, int(z["wseed"])), batch)
    x_t, h_t, prot_x, t = noised_inputs(cfg, batch, z, int(z["T"]))
    eh, ex = eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=0.0, seed=1)
    eng.set_train_precision("bf16")
    with pytest.raises(Exception):
        eng.train_backward(torch.zeros_like(eh), torch.zeros_like(ex))
    with pytest.raises(ValueError):
        eng.set_train_precision("fp8")


def test_bf16_leg_loss_curve_tracks_fp32():
    """200 optimiser steps (FlatAdam, dropout 0.1, a new noise draw and t every step from the same seeds) in both precisions: the
    bf16 leg's loss, averaged over windows of 20 steps, stays within 2 % of the fp32 path's, and both go down."""
    import pharmacoforge_amd as pfa
    from pharmacoforge_amd import synthetic
    B, n_prot, T = 32, 128, 100
    dyn = dict(vector_size=16, n_convs=2, n_hidden_scalars=128, message_norm='mean', dropout=0.1, ff_k=0, pf_k=5,
               n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4)
    pockets = [synthetic.synthetic_pocket(50 + i, n_prot) for i in range(B)]
    sizes = [4 + (i % 5) for i in range(B)]
    curves = {}
    for mode in ("f32", "bf16"):
        m = pfa.PharmacophoreDiff(6, 11, pfa.analysis.ph_idx_to_type, None, n_timesteps=T,
                                  graph_config={'graph_cutoffs': {'pp': 3.5, 'pf': 8, 'fp': 8, 'ff': 9}}, dynamics_config=dyn,
                                  precision=1e-5, lr_scheduler_config={'base_lr': 1e-3, 'weight_decay': 1e-12})
        sd = dict(synthetic.make_state_dict(0))
        sd["gamma.gamma"] = m.state_dict()["gamma.gamma"]
        m.load_state_dict(sd, strict=True)
        m = m.to("cuda").train()
        m.dynamics.set_train_precision(mode)
        eng = m.dynamics.engine()
        assert eng.train_precision() == mode
        gen = torch.Generator().manual_seed(11)
        prot_x, prot_h = torch.cat([p[0] for p in pockets]), torch.cat([p[1] for p in pockets])
        prot_ptr = torch.arange(B + 1, dtype=torch.int64) * n_prot
        pharm_ptr = torch.tensor([0] + list(__import__("itertools").accumulate(sizes)), dtype=torch.int64)
        pp_src, pp_dst = eng.build_pp_edges(prot_x.to("cuda"), prot_ptr)
        Nf = int(pharm_ptr[-1])
        x0 = torch.cat([pockets[i][0].mean(0, keepdim=True) + 2.0 * torch.randn(sizes[i], 3, generator=gen) for i in range(B)])
        h0 = torch.nn.functional.one_hot(torch.randint(0, 6, (Nf,), generator=gen), 6).float()
        g = pfa.PocketGraph(prot_x, prot_h, prot_ptr, pharm_ptr, pp_src, pp_dst, pharm_x0=x0, pharm_h0=h0).to("cuda")
        opt = pfa.FlatAdam(m.dynamics, lr=1e-3, weight_decay=1e-12)
        torch.manual_seed(1234)
        losses = []
        for _ in range(200):
            opt.zero_grad()
            loss = m.training_step(g, 0)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        curves[mode] = torch.stack(losses).cpu().double()
        assert bool(torch.isfinite(curves[mode]).all())
    w32, w16 = curves["f32"].reshape(10, 20).mean(1), curves["bf16"].reshape(10, 20).mean(1)
    assert float(w32[-1]) < 0.9 * float(w32[0]) and float(w16[-1]) < 0.9 * float(w16[0]), (w32, w16)
    rel = ((w16 - w32).abs() / w32).max()
    assert float(rel) <= 0.02, (w32, w16)
