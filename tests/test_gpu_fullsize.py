"""BASELINE.json configurations at FULL size under the library's DEFAULT launch policy (no PFDYN_* switches): which
kernel family, row-group width and work-list form run depends on the batch totals (pf_host.cpp: rg_mode), so the small
golden cases alone do not exercise what a production batch executes.

  config 3   B=128 ragged graphs (3-8 centers), 256-atom pockets: three denoising steps against the CPU oracle with
             shared noise; the policy must have picked 8 rows per wave and the static hoist.
  config 5   B=256 training step, dropout 0.1: every parameter gradient against the oracle's autograd.  The oracle runs
             the batch in chunks of 32 graphs and the chunk gradients are added -- graphs are independent and the loss
             is a sum over centers with a fixed denominator, so the batch gradient IS that sum (linearity; keeps the
             CPU side at ~2 GB instead of ~20 GB of autograd state).
  config 4   a 1-GPU slice (64 pockets x 30 samples, sizes 3..8, max_batch_size 128, the whole T=500 schedule through
             PharmacophoreDiff.sample): counts, finiteness, and the 2-way sharded run reproducing the 1-rank samples BITWISE.

Tolerances: fp32 on both sides; steps 1e-3 (three compounded steps at |x| of a few A), gradients 2e-3 * max|ref| per
tensor."""
import pytest
import torch

import pharmacoforge_amd as pfa
from oracle import pf_oracle as O

pytestmark = pytest.mark.gpu


def _engine(cfg, sd):
    eng = pfa.PfEngine(pharm_nf=cfg.pharm_nf, rec_nf=cfg.rec_nf, n_convs=cfg.n_convs, n_message_gvps=cfg.n_message_gvps,
                       n_update_gvps=cfg.n_update_gvps, n_noise_gvps=cfg.n_noise_gvps, message_norm=cfg.message_norm,
                       ff_k=cfg.ff_k, pf_k=cfg.pf_k,
                       graph_cutoffs={"pp": cfg.cutoff_pp, "pf": cfg.cutoff_pf, "fp": cfg.cutoff_fp, "ff": cfg.cutoff_ff})
    eng.load_state_dict(sd)
    return eng


def _no_policy_overrides():
    import os
    assert not [k for k in os.environ if k.startswith("PFDYN_") and k != "PFDYN_LIB"], "these tests check the DEFAULT launch policy"      # (PFDYN_LIB: which build of the library, not a policy)


def test_config2_batch32_steps_vs_oracle_under_the_default_policy():
    """BASELINE config 2 at its own size (32 x (256 atoms + 6 centers), T = 500) under the DEFAULT launch policy -- the
    configuration bench.py times: three steps at both ends of the schedule against the oracle with shared noise, and the
    kernel forms the policy picked are the ones the bench line reports (conv layers on the n16 kernels, conv layer 0 from
    the static hoist's type tables)."""
    _no_policy_overrides()
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    B = 32
    batch = O.synthetic_batch(range(1000, 1000 + B), 256, [6] * B, cfg)
    T, n = 500, 3
    Nf = int(batch.pharm_ptr[-1])
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(42))
    eng = _engine(cfg, sd)
    eng.set_batch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst)
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    x0, h0 = eng.sample(eng.coef_array(coef, reversed(range(n))), n, noise)          # s = 2, 1, 0
    assert (eng.kernel_family(0), eng.kernel_family(1), eng.l0_hoist()) == (16, 17, 16)
    assert eng.kernel_family(2) == 2 and eng.xchg_timeouts() == 0             # the step ends in the merged node + head / update + build launch
    ne = eng.work()[2]
    assert ne[1] == 5 * Nf and ne[2] == ne[1] and ne[3] == batch.pp_src.numel()
    bidx = batch.batch_idxs()
    init_com = O.segment_mean(batch.prot_x, batch.prot_ptr)
    px = batch.prot_x - init_com[bidx["prot"]]
    x_t, h_t = noise[0][:, :3].clone(), noise[0][:, 3:].clone()
    with torch.no_grad():
        for i, s in enumerate(reversed(range(n))):
            px, x_t, h_t = O.sample_step(sd, cfg, batch, coef, s, px, x_t, h_t, noise[1 + i][:, :3], noise[1 + i][:, 3:])
    ox = x_t - O.segment_mean(px, batch.prot_ptr)[bidx["pharm"]] + init_com[bidx["pharm"]]
    torch.testing.assert_close(x0.cpu(), ox, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(h0.cpu(), h_t, rtol=1e-3, atol=1e-3)
    x1, h1 = eng.sample(eng.coef_array(coef, reversed(range(T))), n, noise)           # s = 499, 498, 497
    ox1, oh1 = O.sample_given_receptor(sd, cfg, batch, T, 1e-5, noise, n_steps=n)
    torch.testing.assert_close(x1.cpu(), ox1, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(h1.cpu(), oh1, rtol=1e-3, atol=1e-3)
    # a single dynamics call of that batch at the tolerance of a call (2e-4), per-graph timesteps (the non-shared tables)
    gen = torch.Generator().manual_seed(4)
    xt, ht, t = 2.0 * torch.randn(Nf, 3, generator=gen), torch.randn(Nf, 6, generator=gen), torch.rand(B, generator=gen)
    eh, ex = eng.dynamics(xt, ht, t, prot_x=batch.prot_x)
    oh, oxx = O.dynamics_forward(sd, cfg, batch, batch.prot_x, xt, ht, t)
    torch.testing.assert_close(eh.cpu(), oh, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(ex.cpu(), oxx, rtol=2e-4, atol=2e-4)


def test_config3_ragged_batch128_steps_vs_oracle():
    _no_policy_overrides()
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    B = 128
    sizes = [3 + (i % 6) for i in range(B)]
    batch = O.synthetic_batch(range(500, 500 + B), 256, sizes, cfg)
    T, n = 500, 3
    Nf = int(batch.pharm_ptr[-1])
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(42))
    eng = _engine(cfg, sd)
    eng.set_batch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst)
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    # the LAST steps of the schedule (s = 2, 1, 0): small noise, the centers stay inside the pocket
    x0, h0 = eng.sample(eng.coef_array(coef, reversed(range(n))), n, noise)
    assert eng.kernel_family(0) == 8 and eng.l0_hoist() > 0, (eng.kernel_family(0), eng.l0_hoist())
    ne = eng.work()[2]
    assert ne[1] == 5 * Nf and ne[2] == ne[1] and ne[3] == batch.pp_src.numel()
    # oracle: same steps (its loop walks s = T-1 ...; drive the steps directly)
    bidx = batch.batch_idxs()
    init_com = O.segment_mean(batch.prot_x, batch.prot_ptr)
    px = batch.prot_x - init_com[bidx["prot"]]
    x_t, h_t = noise[0][:, :3].clone(), noise[0][:, 3:].clone()
    with torch.no_grad():
        for i, s in enumerate(reversed(range(n))):
            px, x_t, h_t = O.sample_step(sd, cfg, batch, coef, s, px, x_t, h_t, noise[1 + i][:, :3], noise[1 + i][:, 3:])
    ox = x_t - O.segment_mean(px, batch.prot_ptr)[bidx["pharm"]] + init_com[bidx["pharm"]]
    torch.testing.assert_close(x0.cpu(), ox, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(h0.cpu(), h_t, rtol=1e-3, atol=1e-3)
    # and the high-noise head of the schedule (s = 499 ...), where 1/alpha_t|s = 1.6 stretches the coordinates
    x1, h1 = eng.sample(eng.coef_array(coef, reversed(range(T))), n, noise)
    ox1, oh1 = O.sample_given_receptor(sd, cfg, batch, T, 1e-5, noise, n_steps=n)
    torch.testing.assert_close(x1.cpu(), ox1, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(h1.cpu(), oh1, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("arch", ["dev", "radius_pf"])
def test_pockets_above_512_atoms_vs_oracle(arch):
    """Pockets beyond the fast forms' gate (the latency-optimised update + build, the fused and the merged launch all need
    max_np <= 512: pf_host.cpp LaunchPolicy; dynamics_gvp.py:187-227 has no such bound): a batch of 700-, 540-, 96- and 820-atom
    pockets takes the general build, and must give the oracle's steps and exactly its edges -- three denoising steps at both
    ends of the schedule, kNN pf edges (dev.yml) and radius pf edges (the class defaults' pf_k = 0), default launch policy
    (VERDICT r4 missing #4: no test covered more than 475 atoms)."""
    _no_policy_overrides()
    cfg = O.DynamicsConfig() if arch == "dev" else O.DynamicsConfig(pf_k=0, message_norm=1)
    sd = O.make_state_dict(cfg, 0)
    n_prot, sizes = [700, 540, 96, 820], [6, 8, 3, 5]
    batch = O.synthetic_batch([2000 + i for i in range(4)], n_prot, sizes, cfg)
    assert int(batch.prot_ptr[-1]) == sum(n_prot)
    T, n = 500, 3
    Nf = int(batch.pharm_ptr[-1])
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(7))
    eng = _engine(cfg, sd)
    eng.set_batch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst)
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    x0, h0 = eng.sample(eng.coef_array(coef, reversed(range(n))), n, noise)          # s = 2, 1, 0
    assert eng.kernel_family(cfg.n_convs) == 0                                        # neither a tail nor the merged launch: the general build ran
    bidx = batch.batch_idxs()
    init_com = O.segment_mean(batch.prot_x, batch.prot_ptr)
    px = batch.prot_x - init_com[bidx["prot"]]
    x_t, h_t = noise[0][:, :3].clone(), noise[0][:, 3:].clone()
    with torch.no_grad():
        for i, s in enumerate(reversed(range(n))):
            px, x_t, h_t = O.sample_step(sd, cfg, batch, coef, s, px, x_t, h_t, noise[1 + i][:, :3], noise[1 + i][:, 3:])
    ox = x_t - O.segment_mean(px, batch.prot_ptr)[bidx["pharm"]] + init_com[bidx["pharm"]]
    torch.testing.assert_close(x0.cpu(), ox, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(h0.cpu(), h_t, rtol=1e-3, atol=1e-3)
    # the edges the LAST step's build emitted for the next call, against the oracle's on its own final state: exact sets
    edges = O.build_dynamic_edges(cfg, batch, px, x_t)
    for et, name in enumerate(("ff", "pf", "fp")):
        s_, d_ = eng.get_edges(et)
        os_, od_ = edges[name]
        assert set(zip(s_.tolist(), d_.tolist())) == set(zip(os_.tolist(), od_.tolist())), (arch, name)
    x1, h1 = eng.sample(eng.coef_array(coef, reversed(range(T))), n, noise)           # s = 499, 498, 497
    ox1, oh1 = O.sample_given_receptor(sd, cfg, batch, T, 1e-5, noise, n_steps=n)
    torch.testing.assert_close(x1.cpu(), ox1, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(h1.cpu(), oh1, rtol=1e-3, atol=1e-3)
    gen = torch.Generator().manual_seed(4)
    xt, ht, t = 2.0 * torch.randn(Nf, 3, generator=gen), torch.randn(Nf, 6, generator=gen), torch.rand(4, generator=gen)
    eh, ex = eng.dynamics(xt, ht, t, prot_x=batch.prot_x)
    oh, oxx = O.dynamics_forward(sd, cfg, batch, batch.prot_x, xt, ht, t)
    torch.testing.assert_close(eh.cpu(), oh, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(ex.cpu(), oxx, rtol=2e-4, atol=2e-4)


_CONFIG5 = {}


def _config5_case():
    """Config 5's batch (256 pockets x 256 atoms, 4-8 centers, dropout 0.1), its inputs, the engine's dropout masks, and the
    oracle's outputs and parameter gradients (autograd on the CPU, 32 graphs at a time) -- computed once, used by the fp32 test
    and by the bf16 leg's contract test."""
    if _CONFIG5:
        return _CONFIG5
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    B, CH = 256, 32
    sizes = [4 + (i % 5) for i in range(B)]                                  # 4..8 centers (dev.yml subsampling range)
    batch = O.synthetic_batch(range(900, 900 + B), 256, sizes, cfg)
    Np, Nf = int(batch.prot_ptr[-1]), int(batch.pharm_ptr[-1])
    gen = torch.Generator().manual_seed(5)
    bidx = batch.batch_idxs()
    com = O.segment_mean(batch.prot_x, batch.prot_ptr)
    x_t = com[bidx["pharm"]] * 0 + 2.5 * torch.randn(Nf, 3, generator=gen)
    prot_x = batch.prot_x - com[bidx["prot"]]
    h_t = torch.randn(Nf, cfg.pharm_nf, generator=gen)
    t = torch.rand(B, generator=gen)
    w_h, w_x = torch.randn(Nf, cfg.pharm_nf, generator=gen) / Nf, torch.randn(Nf, 3, generator=gen) / Nf
    p_drop, seed = 0.1, 4242
    eng = _engine(cfg, sd)
    eng.set_batch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst)
    masks = [(eng.dropout_mask(l, 0, p_drop, seed).cpu(), eng.dropout_mask(l, 1, p_drop, seed).cpu()) for l in range(cfg.n_convs)]
    ref = {k: torch.zeros_like(v) for k, v in sd.items()}
    oh_all, ox_all = [], []
    for g0 in range(0, B, CH):
        g1 = g0 + CH
        p0, p1, f0, f1 = int(batch.prot_ptr[g0]), int(batch.prot_ptr[g1]), int(batch.pharm_ptr[g0]), int(batch.pharm_ptr[g1])
        em = (batch.pp_dst >= p0) & (batch.pp_dst < p1)
        sub = O.PocketBatch(batch.prot_x[p0:p1], batch.prot_h[p0:p1], batch.prot_ptr[g0:g1 + 1] - p0,
                            batch.pharm_ptr[g0:g1 + 1] - f0, batch.pp_src[em] - p0, batch.pp_dst[em] - p0)
        drop = []
        for m0, m1 in masks:
            d = {}
            for nt, sl in (("prot", slice(p0, p1)), ("pharm", slice(Np + f0, Np + f1))):
                d[nt] = (m0[sl, :128], m0[sl, 128:], m1[sl, :128], m1[sl, 128:])
            drop.append(d)
        leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        with torch.enable_grad():
            oh, ox = O.dynamics_forward(leaf, cfg, sub, prot_x[p0:p1], x_t[f0:f1], h_t[f0:f1], t[g0:g1], dropout=drop)
            ((oh * w_h[f0:f1]).sum() + (ox * w_x[f0:f1]).sum()).backward()
        for k, v in leaf.items():
            if v.grad is not None:
                ref[k] += v.grad
        oh_all.append(oh.detach()); ox_all.append(ox.detach())
    _CONFIG5.update(eng=eng, x_t=x_t, h_t=h_t, t=t, prot_x=prot_x, w_h=w_h, w_x=w_x, p_drop=p_drop, seed=seed, ref=ref,
                    oh=torch.cat(oh_all), ox=torch.cat(ox_all))
    return _CONFIG5


def test_config5_training_step_batch256_gradients_vs_oracle():
    _no_policy_overrides()
    c = _config5_case()
    eng, ref, oh_all, ox_all = c["eng"], c["ref"], c["oh"], c["ox"]
    eng.set_train_precision("f32")
    eps_h, eps_x = eng.train_forward(c["x_t"], c["h_t"], c["t"], prot_x=c["prot_x"], dropout=c["p_drop"], seed=c["seed"])
    grad = eng.train_backward(c["w_h"], c["w_x"]).cpu()
    assert float((eps_h.cpu() - oh_all).abs().max()) < 5e-4 * max(1.0, float(oh_all.abs().max()))
    assert float((eps_x.cpu() - ox_all).abs().max()) < 5e-4 * max(1.0, float(ox_all.abs().max()))
    bad, live = [], 0
    for name, off, n in eng.param_layout():
        r = ref[name].reshape(-1)
        if n == 0:
            continue
        scale = float(r.abs().max())
        live += scale > 0
        err = float((grad[off:off + n] - r).abs().max())
        if err > 2e-3 * scale + 1e-9:
            bad.append((name, err, scale))
    assert not bad, (bad[:8], len(bad))
    assert live >= 150


def test_config5_bf16_leg_batch256_gradients_vs_oracle():
    """BASELINE config 5's bf16 leg (a labelled secondary line: the reference trains in fp32, pharmacodiff.py:162-263) against the
    fp32 ORACLE's autograd at the full batch: every parameter tensor's gradient with cosine >= 0.999 and relative L2 error
    <= 2e-2 (measured: >= 0.99993 / <= 1.3e-2), outputs within 5e-3 relative L2."""
    _no_policy_overrides()
    c = _config5_case()
    eng, ref, oh_all, ox_all = c["eng"], c["ref"], c["oh"], c["ox"]
    eng.set_train_precision("bf16")
    try:
        eps_h, eps_x = eng.train_forward(c["x_t"], c["h_t"], c["t"], prot_x=c["prot_x"], dropout=c["p_drop"], seed=c["seed"])
        grad = eng.train_backward(c["w_h"], c["w_x"]).cpu()
    finally:
        eng.set_train_precision("f32")
    assert float((eps_h.cpu() - oh_all).norm() / oh_all.norm()) <= 5e-3
    assert float((eps_x.cpu() - ox_all).norm() / ox_all.norm()) <= 5e-3
    live, worst = 0, (1.0, 0.0)
    for name, off, n in eng.param_layout():
        r = ref[name].reshape(-1).double()
        if n == 0 or float(r.abs().max()) == 0.0:
            continue
        g = grad[off:off + n].double()
        cos = float((r * g).sum() / (r.norm() * g.norm()))
        rel = float((r - g).norm() / r.norm())
        live += 1
        worst = (min(worst[0], cos), max(worst[1], rel))
        assert cos >= 0.999 and rel <= 2e-2, (name, cos, rel)
    assert live >= 150, live


def _slice_model(T):
    m = pfa.PharmacophoreDiff(6, 11, pfa.analysis.ph_idx_to_type, None, n_timesteps=T,
                              graph_config={'graph_cutoffs': {'pp': 3.5, 'pf': 8, 'fp': 8, 'ff': 9}},
                              dynamics_config=dict(vector_size=16, n_convs=2, n_hidden_scalars=128, message_norm='mean', dropout=0.1,
                                                   ff_k=0, pf_k=5, n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4), precision=1e-5)
    sd = dict(O.make_state_dict(O.DynamicsConfig(), 0))
    sd["gamma.gamma"] = m.state_dict()["gamma.gamma"]
    m.load_state_dict(sd, strict=True)
    return m.to("cuda").eval()


def test_config4_slice_sharded_equals_single_rank_bitwise():
    _no_policy_overrides()
    cfg = O.DynamicsConfig()
    T, n_pockets, per = 500, 64, 30                   # 1,920 graphs = 15 full batches of 128
    m = _slice_model(T)
    sizes_cycle = [3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 7, 7, 7, 7, 7, 8, 8, 8, 8, 8]     # README pattern
    pockets = []
    for i in range(n_pockets):                                  # pocket sizes spread 3x: 160..475 atoms
        b = O.synthetic_batch([3000 + i], 160 + 5 * i, 1, cfg)
        pockets.append(pfa.PocketGraph(b.prot_x, b.prot_h, b.prot_ptr, b.pharm_ptr, b.pp_src, b.pp_dst,
                                       torch.zeros(1, 3), torch.zeros(1, 6)))
    n_pharms = [list(sizes_cycle[:per]) for _ in range(n_pockets)]
    torch.manual_seed(11)
    full = m.sample(pockets, n_pharms, max_batch_size=128)
    assert [len(o) for o in full] == [per] * n_pockets
    assert [[p.n_ph_centers for p in o] for o in full] == n_pharms
    assert all(torch.isfinite(p.ph_coords).all() for o in full for p in o)
    assert m.dynamics.engine().kernel_family(0) in (4, 8)
    # batches in flight on several HIP streams / handles: a batch's result does not depend on the lane it ran on
    for lanes in (1, 3):
        torch.manual_seed(11)
        again = m.sample(pockets, n_pharms, max_batch_size=128, lanes=lanes)
        for i in range(n_pockets):
            for p, q in zip(full[i], again[i]):
                assert torch.equal(p.ph_coords, q.ph_coords) and torch.equal(p.ph_feats_idxs, q.ph_feats_idxs), (lanes, i)
    halves = []
    for r in range(2):
        torch.manual_seed(11)                                   # the ranks of a job share the seed
        halves.append(m.sample(pockets, n_pharms, max_batch_size=128, rank=r, world_size=2))
    n0 = sum(len(o) for o in halves[0]); n1 = sum(len(o) for o in halves[1])
    assert n0 + n1 == n_pockets * per and min(n0, n1) >= 0.4 * (n0 + n1)
    for i in range(n_pockets):
        merged = {}
        for h in halves:
            for p in h[i]:
                merged.setdefault(p.n_ph_centers, []).append(p)
        assert sum(len(v) for v in merged.values()) == per
        got = sorted((p.ph_coords.flatten().tolist(), p.ph_feats_idxs.tolist()) for v in merged.values() for p in v)
        want = sorted((p.ph_coords.flatten().tolist(), p.ph_feats_idxs.tolist()) for p in full[i])
        assert got == want, f"pocket {i}: sharded samples differ from the single-rank ones"


def test_workspace_reuse_leaves_no_trace_of_the_previous_batch():
    """pf_set_pocket_batch keeps its allocations across batches and clears only what a batch reads before writing (the
    zero message row, the counters): results on a batch must not depend on what the handle ran before, bit for bit --
    inference and training alike."""
    _no_policy_overrides()
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    big = O.synthetic_batch(range(40, 72), 200, [3 + (i % 6) for i in range(32)], cfg)
    small = O.synthetic_batch(range(80, 85), 120, [4, 7, 3, 8, 5], cfg)
    T, n = 100, 12
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    gen = torch.Generator().manual_seed(1)
    nz_big = torch.randn(n + 1, int(big.pharm_ptr[-1]), 9, generator=gen)
    nz_small = torch.randn(n + 1, int(small.pharm_ptr[-1]), 9, generator=gen)

    def bind(eng, b):
        eng.set_batch(b.prot_x, b.prot_h, b.prot_ptr, b.pharm_ptr, b.pp_src, b.pp_dst)

    fresh = _engine(cfg, sd)
    bind(fresh, small)
    arr = fresh.coef_array(coef, reversed(range(n)))
    ref_x, ref_h = fresh.sample(arr, n, nz_small)
    used = _engine(cfg, sd)
    bind(used, big)
    used.sample(arr, n, nz_big)
    bind(used, small)                                            # same allocation, smaller batch
    x, h = used.sample(arr, n, nz_small)
    assert torch.equal(x, ref_x) and torch.equal(h, ref_h)
    # training: gradients on `small` after a training step on `big`
    Nf, B = int(small.pharm_ptr[-1]), small.batch_size
    x_t, h_t, t = torch.randn(Nf, 3, generator=gen), torch.randn(Nf, 6, generator=gen), torch.rand(B, generator=gen)
    w_h, w_x = torch.randn(Nf, 6, generator=gen), torch.randn(Nf, 3, generator=gen)
    fresh.train_forward(x_t, h_t, t, prot_x=small.prot_x, dropout=0.1, seed=9)
    g_ref = fresh.train_backward(w_h, w_x)
    Nfb = int(big.pharm_ptr[-1])
    bind(used, big)
    used.train_forward(torch.randn(Nfb, 3, generator=gen), torch.randn(Nfb, 6, generator=gen), torch.rand(32, generator=gen),
                       prot_x=big.prot_x, dropout=0.1, seed=3)
    used.train_backward(torch.randn(Nfb, 6, generator=gen), torch.randn(Nfb, 3, generator=gen))
    bind(used, small)
    used.train_forward(x_t, h_t, t, prot_x=small.prot_x, dropout=0.1, seed=9)
    g = used.train_backward(w_h, w_x)
    assert torch.equal(g, g_ref)


def _copies_batch(cfg, pockets, sizes_per_pocket):
    """[(seed, n_prot)] x [[sizes]] -> (PocketBatch of the copies in pocket order, pocket uid per graph)."""
    parts, uid = [], []
    for k, ((seed, n), sizes) in enumerate(zip(pockets, sizes_per_pocket)):
        pocket = O.synthetic_batch([seed], n, 1, cfg)
        parts += [O.copy_pocket(pocket, s) for s in sizes]
        uid += [k + 1] * len(sizes)
    return O.concat_pockets(parts), torch.tensor(uid)


def test_pocket_sharing_equals_the_per_copy_path():
    """Copies of one pocket at the same timestep share conv layer 0's protein->protein messages (pf_set_pocket_groups):
    a 40-step trajectory of 3 pockets x (24, 20, 21) ragged copies with sharing == without, against the oracle too; the
    shared launch computes fewer edges; per-graph timesteps (pf_dynamics_forward) on the same bind fall back to the
    per-copy path; a false claim is refused."""
    _no_policy_overrides()
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    sizes = [3, 8, 5, 4, 6, 7, 3, 5, 4, 8, 3, 6, 6, 3, 4, 8, 5, 7, 3, 4, 6, 5, 7, 8]
    batch, uid = _copies_batch(cfg, [(501, 120), (502, 200), (503, 150)], [sizes, sizes[:20], sizes[3:]])
    T, n = 100, 40
    Nf = int(batch.pharm_ptr[-1])
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(3))
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    res, work = [], []
    for shared in (False, True):
        eng = _engine(cfg, sd)
        eng.set_batch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst,
                      pocket_uid=uid if shared else None)
        arr = eng.coef_array(coef, reversed(range(n)))
        x, h = eng.sample(arr, n, noise)
        res.append((x.cpu(), h.cpu()))
        work.append(eng.work_detail())
        if shared:                                   # per-graph t on the same bind: the per-copy path, against the oracle
            gen = torch.Generator().manual_seed(4)
            x_t, h_t, t = 2.0 * torch.randn(Nf, 3, generator=gen), torch.randn(Nf, 6, generator=gen), torch.rand(batch.batch_size, generator=gen)
            eh, ex = eng.dynamics(x_t, h_t, t, prot_x=batch.prot_x)
            oh, ox = O.dynamics_forward(sd, cfg, batch, batch.prot_x, x_t, h_t, t)
            torch.testing.assert_close(eh.cpu(), oh, rtol=2e-4, atol=2e-4)
            torch.testing.assert_close(ex.cpu(), ox, rtol=2e-4, atol=2e-4)
            x2, h2 = eng.sample(arr, n, noise)       # and back to the shared form: bitwise what it gave before
            assert torch.equal(x2.cpu(), res[1][0]) and torch.equal(h2.cpu(), res[1][1])
    torch.testing.assert_close(res[1][0], res[0][0], rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(res[1][1], res[0][1], rtol=2e-4, atol=2e-4)
    assert work[1]["executed_edges_per_layer"][0] < 0.8 * work[0]["executed_edges_per_layer"][0], (work[0], work[1])
    assert work[1]["flops"] == work[0]["flops"]                    # the reference-equivalent work is the same
    # a claim that does not hold: graph 1 named a copy of a pocket with another atom count
    eng = _engine(cfg, sd)
    bad = uid.clone(); bad[24] = 1
    with pytest.raises(pfa.PfError, match="not a copy"):
        eng.set_batch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst, pocket_uid=bad)
    # ... and one that the host copy of the coordinates exposes
    px = batch.prot_x.clone(); px[130] += 0.01                     # an atom of the second copy of the first pocket
    with pytest.raises(pfa.PfError, match="coordinates / features differ"):
        eng.set_batch(px, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst, pocket_uid=uid)


def test_config2_whole_T500_reverse_process_vs_oracle():
    """BASELINE config 2 exactly as stated -- 256-atom pockets, 6 centers, batch 32, ALL T = 500 steps of the reverse process
    (pharmacodiff.py:433-514: the loop :466-472) -- against the oracle with shared noise, under the default launch policy: every
    25th frame of the trajectory and the end point at 2e-2 A ABSOLUTE (features: 2e-2), in the bounded regime of
    tests/golden/traj_c1_T500_bounded.npz (schedule precision 0.25: 1 / alpha_T <= 2, the centers stay inside the pockets, so
    ff / pf / fp edges exist at every step and the tolerance needs no relative part).  ~2 min of CPU for the oracle."""
    _no_policy_overrides()
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    B, T, prec = 32, 500, 0.25
    batch = O.synthetic_batch(range(1000, 1000 + B), 256, [6] * B, cfg)
    Nf = int(batch.pharm_ptr[-1])
    noise = torch.randn(T + 1, Nf, 9, generator=torch.Generator().manual_seed(42))
    eng = _engine(cfg, sd)
    eng.set_batch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst)
    coef = O.step_coefficients(O.gamma_table(T, prec), T)
    res = eng.sample(eng.coef_array(coef, reversed(range(T))), T, noise, trajectory=True)
    assert (eng.kernel_family(0), eng.kernel_family(1), eng.l0_hoist()) == (16, 17, 16)
    assert eng.kernel_family(2) == 2 and eng.xchg_timeouts() == 0
    ne = eng.work()[2]                                                        # edges of the last dynamics call (step s = 0)
    # (most ordered pairs of centers are within the 9 A ff cutoff at the end; every center keeps its 5 nearest atoms)
    assert 0.85 * B * 6 * 5 <= ne[0] <= B * 6 * 5 and ne[1] == 5 * Nf and ne[2] == ne[1] and ne[3] == batch.pp_src.numel()
    nthr = torch.get_num_threads()
    torch.set_num_threads(16)                 # (the oracle's best thread count on the GPU boxes; their default oversubscribes: 100x slower)
    try:
        with torch.no_grad():
            ox, oh, oframes = O.sample_given_receptor(sd, cfg, batch, T, prec, noise, return_traj=True)
    finally:
        torch.set_num_threads(nthr)
    opos = torch.stack([f[0] for f in oframes]); ofeat = torch.stack([f[1] for f in oframes])
    assert opos.shape[0] == T + 1
    com = O.segment_mean(batch.prot_x, batch.prot_ptr)[batch.batch_idxs()["pharm"]]
    assert float((opos - com).norm(dim=-1).max()) < 10.7                      # the centers never leave the pockets (radius 10.7 A)
    sel = list(range(0, T + 1, 25))
    worst = float((res[2].cpu()[sel] - opos[sel]).abs().max())
    torch.testing.assert_close(res[2].cpu()[sel], opos[sel], rtol=0.0, atol=2e-2)
    torch.testing.assert_close(res[3].cpu()[sel], ofeat[sel], rtol=0.0, atol=2e-2)
    torch.testing.assert_close(res[0].cpu(), ox, rtol=0.0, atol=2e-2)
    torch.testing.assert_close(res[1].cpu(), oh, rtol=0.0, atol=2e-2)
    assert worst < 2e-2, worst
