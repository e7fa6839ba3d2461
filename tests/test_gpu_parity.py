"""GPU parity tests proper: the HIP path (through the C ABI of include/pfdyn.h) against the CPU
oracle on the same seeded inputs and against the golden vectors recorded from the reference.

Tolerances (fp32 arithmetic on both sides; the HIP kernels accumulate dot products in MFMA
k-order, the oracle in BLAS order):
    single dynamics call / conv layer :  |err| <= 2e-4 + 2e-4*|ref|
    T-step stochastic trajectory       :  |err| <= 5e-3 + 5e-3*|ref|   (rounding compounds over T steps)
Edge sets must match exactly (bit-exact integer work)."""
import pytest
import torch

from oracle import pf_oracle as O
from helpers import DYN_CASES, batch_from, edge_set, load

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-4, 2e-4


def engine_for(cfg: O.DynamicsConfig, sd):
    import pharmacoforge_amd as pfa
    eng = pfa.PfEngine(pharm_nf=cfg.pharm_nf, rec_nf=cfg.rec_nf, n_convs=cfg.n_convs,
                       n_message_gvps=cfg.n_message_gvps, n_update_gvps=cfg.n_update_gvps,
                       n_noise_gvps=cfg.n_noise_gvps, message_norm=cfg.message_norm, ff_k=cfg.ff_k, pf_k=cfg.pf_k,
                       graph_cutoffs={"pp": cfg.cutoff_pp, "pf": cfg.cutoff_pf, "fp": cfg.cutoff_fp, "ff": cfg.cutoff_ff})
    eng.load_state_dict(sd)
    return eng


def set_batch(eng, batch: O.PocketBatch, prot_x=None):
    eng.set_batch(batch.prot_x if prot_x is None else prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr,
                  batch.pp_src, batch.pp_dst)


def close(a, b, rtol=RTOL, atol=ATOL):
    torch.testing.assert_close(a.cpu(), b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", list(DYN_CASES))
def test_dynamics_vs_golden_and_oracle(name):
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    eng = engine_for(cfg, sd)
    set_batch(eng, batch, z["prot_x"])
    eps_h, eps_x = eng.dynamics(z["x_t"], z["h_t"], z["t"])
    # edge sets of that call == the reference's (golden) == the oracle's
    for i, et in enumerate(O.ETYPES):
        s, d = eng.get_edges(i)
        assert edge_set(s, d) == edge_set(z[f"e_{et}_src"].long(), z[f"e_{et}_dst"].long()), et
        assert s.numel() == z[f"e_{et}_src"].numel()
    close(eps_h, z["eps_h"]); close(eps_x, z["eps_x"])
    oh, ox = O.dynamics_forward(sd, cfg, batch, z["prot_x"], z["x_t"], z["h_t"], z["t"])
    close(eps_h, oh); close(eps_x, ox)


@pytest.mark.parametrize("name", list(DYN_CASES))
def test_conv_layer_nonzero_vectors(name):
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    eng = engine_for(cfg, sd)
    set_batch(eng, batch, z["prot_x"])
    li = int(z["conv_layer_index"])
    hp, vp, hf, vf = eng.conv_layer(li, z["prot_x"], z["x_t"], z["conv_in_h_prot"], z["conv_in_v_prot"],
                                    z["conv_in_h_pharm"], z["conv_in_v_pharm"])
    close(hp, z["conv_out_h_prot"]); close(vp, z["conv_out_v_prot"])
    close(hf, z["conv_out_h_pharm"]); close(vf, z["conv_out_v_pharm"])


TAIL_FORMS = {"rg": 4, "n16": 16, "merged": 2, "off": 0}      # form of a step's end -> pf_debug_kernel_family(n_convs)


def set_tail(monkeypatch, tail):
    """The end of a denoising step: the merged launch (the default policy: node + head items and every graph's update + build as
    workgroups of one grid, k_rg_node_hs_build), one tail launch per graph (row-group or n16 form), or separate launches."""
    if tail in ("rg", "n16"):
        monkeypatch.setenv("PFDYN_N16", "15")
        monkeypatch.setenv("PFDYN_TAIL_FORM", tail)
    elif tail == "off":
        monkeypatch.setenv("PFDYN_HS_BUILD", "0")


@pytest.mark.parametrize("tail", list(TAIL_FORMS))
@pytest.mark.parametrize("name,ep", [("traj_c1.npz", False), ("traj_ragged.npz", False), ("traj_c1_T500.npz", False),
                                     ("traj_endpoint.npz", True)])
def test_trajectory_vs_golden(name, ep, tail, monkeypatch):
    """pf_sample against the reference's own trajectories (its noise draws injected): config 1 at T=50 and over the whole
    T=500 schedule -- all 501 frames; with seeded random weights the centers drift to |x| ~ 480 A by the end, so the
    relative part of the tolerance carries the late frames --, a ragged batch, and the endpoint parameterisation of
    coordinates and features (pharmacodiff.py:413-420).  Every form of a step's end: the tail launch (node update + noise
    head + sampler update + edge build, one workgroup per graph; row-group form = default, n16 form) and the separate
    node + head and update + build launches (PFDYN_N16 without bit 3)."""
    set_tail(monkeypatch, tail)
    z = load(name)
    cfg = O.DynamicsConfig()
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    T = int(z["T"])
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    arr = eng.coef_array(coef, reversed(range(T)))
    traj = "pos_frames" in z
    res = eng.sample(arr, T, z["noise"], trajectory=traj, ep_coord=ep, ep_feat=ep)
    assert eng.kernel_family(cfg.n_convs) == TAIL_FORMS[tail]
    assert eng.xchg_timeouts() == 0
    close(res[0], z["x0"], 5e-3, 5e-3); close(res[1], z["h0"], 5e-3, 5e-3)
    if traj:
        close(res[2], z["pos_frames"], 5e-3, 5e-3); close(res[3], z["feat_frames"], 5e-3, 5e-3)


@pytest.mark.parametrize("tail", list(TAIL_FORMS))
def test_bounded_T500_trajectory_every_frame_absolute(tail, monkeypatch):
    """The whole T = 500 reverse process in the regime a trained model lives in -- every center inside the pocket at every step,
    ff / pf / fp edges present throughout -- against the reference's own run (tests/golden/traj_c1_T500_bounded.npz: the
    reference's sampler with its `precision` argument at 0.25, which bounds 1 / alpha_T by 2; see make_golden.py for why
    scaled weights cannot do that): all 501 frames with an ABSOLUTE tolerance of 2e-2 A, no relative part; edge sets of the
    last step; and the same through the oracle."""
    set_tail(monkeypatch, tail)
    z = load("traj_c1_T500_bounded.npz")
    cfg = O.DynamicsConfig()
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    T, prec = int(z["T"]), float(z["precision"])
    assert T == 500 and prec == 0.25
    pos = z["pos_frames"]
    com = O.segment_mean(batch.prot_x, batch.prot_ptr)
    assert float((pos - com).norm(dim=-1).max()) < 4.0                       # pocket radius 6.7 A: the centers never leave it
    pd = torch.cdist(pos[-100:], pos[-100:])
    assert float(pd.max()) < cfg.cutoff_ff                                    # => every ordered pair is an ff edge in the last 100 steps
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    coef = O.step_coefficients(O.gamma_table(T, prec), T)
    res = eng.sample(eng.coef_array(coef, reversed(range(T))), T, z["noise"], trajectory=True)
    assert eng.kernel_family(cfg.n_convs) == TAIL_FORMS[tail]
    assert eng.xchg_timeouts() == 0
    for got, ref in ((res[0], z["x0"]), (res[1], z["h0"]), (res[2], z["pos_frames"]), (res[3], z["feat_frames"])):
        torch.testing.assert_close(got.cpu(), ref, rtol=0.0, atol=2e-2)
    Nf = int(batch.pharm_ptr[-1])
    ne = eng.work()[2]                                                        # edges of the last dynamics call (step s = 0)
    assert ne[0] == Nf * (Nf - 1) and ne[1] == 5 * Nf and ne[2] == ne[1]
    worst = float((res[2].cpu() - z["pos_frames"]).abs().max())
    assert worst < 2e-2, worst


UNIT_TOL = 5e-5


def test_chain_units_vs_reference_modules():
    """The chain code of the row-group kernels in isolation (pf_debug_chain: rg_gvp / rg_flush / rg_layernorm on supplied
    rows) against the outputs of the reference's own modules on the same rows (tests/golden/units.npz, recorded by
    make_golden.py from gvp.py:89-116 GVP chains, gvp.py:152-166 GVPLayerNorm incl. an all-zero vector row, dynamics_gvp.py:10-42
    NoisePredictionBlock).  A single GVP of a chain cannot be cut out of the quad stream (block j carries the gates of GVP
    j - 1), so the msg0 / msg1 vectors of the file stay with the oracle test; chains, norm and head are compared here."""
    z = load("units.npz")
    cfg = O.DynamicsConfig()
    eng = engine_for(cfg, O.make_state_dict(cfg, 0))
    so, vo = eng.debug_chain(0, 0, 1, z["chain_s"], z["chain_v"])            # edge_message_fns.prot_pf_pharm, 3 GVPs
    close(so, z["chain_so"], UNIT_TOL, UNIT_TOL); close(vo, z["chain_vo"], UNIT_TOL, UNIT_TOL)
    so, vo = eng.debug_chain(1, 0, 0, z["upd_s"], z["upd_v"])                # node_update_fns.prot, 2 GVPs
    close(so, z["upd_so"], UNIT_TOL, UNIT_TOL); close(vo, z["upd_vo"], UNIT_TOL, UNIT_TOL)
    so, vo = eng.debug_chain(2, 0, 2, z["ln_s"], z["ln_v"])                  # message_layer_norms.pharm
    close(so, z["ln_so"], UNIT_TOL, UNIT_TOL); close(vo, z["ln_vo"], UNIT_TOL, UNIT_TOL)
    assert torch.all(vo[5] == 0) and torch.isfinite(vo).all()
    eh, ex = eng.debug_chain(3, 0, 0, z["head_s"], z["head_v"])              # noise_predictor: 4 GVPs + to_scalar_output
    close(eh, z["head_eh"], UNIT_TOL, UNIT_TOL); close(ex, z["head_ex"], UNIT_TOL, UNIT_TOL)


@pytest.mark.parametrize("arch", ["dev", "deep"])
def test_chain_units_every_chain_vs_oracle(arch):
    """Every message chain (layer x edge type), update chain and norm (layer x node type) and the head of a network through
    pf_debug_chain against the oracle's restatement on random rows, ragged row counts (1..4 rows in the last wave)."""
    cfg = O.DynamicsConfig() if arch == "dev" else O.DynamicsConfig(n_convs=3, n_message_gvps=2, n_update_gvps=3, n_noise_gvps=2)
    sd = O.make_state_dict(cfg, 7)
    eng = engine_for(cfg, sd)
    gen = torch.Generator().manual_seed(3)
    et_names = ["pharm_ff_pharm", "prot_pf_pharm", "pharm_fp_prot", "prot_pp_prot"]
    for layer in range(cfg.n_convs):
        p = f"dynamics.noise_predictor.conv_layers.{layer}."
        for et in range(4):
            n = 5 + 4 * layer + et                                            # 5..: every remainder mod 4 occurs
            s, v = torch.randn(n, 144, generator=gen), torch.randn(n, 17, 3, generator=gen)
            s[:, 128:] = s[:, 128:].abs().clamp(max=1.0)                      # rbf values live in (0, 1]
            so, vo = eng.debug_chain(0, layer, et, s, v)
            ro, rv = O.gvp_chain(sd, p + f"edge_message_fns.{et_names[et]}.", cfg.n_message_gvps, s, v)
            close(so, ro, UNIT_TOL, UNIT_TOL); close(vo, rv, UNIT_TOL, UNIT_TOL)
        for nt, name in enumerate(("prot", "pharm")):
            n = 6 + nt
            s, v = torch.randn(n, 128, generator=gen), torch.randn(n, 16, 3, generator=gen)
            so, vo = eng.debug_chain(1, layer, nt, s, v)
            ro, rv = O.gvp_chain(sd, p + f"node_update_fns.{name}.", cfg.n_update_gvps, s, v)
            close(so, ro, UNIT_TOL, UNIT_TOL); close(vo, rv, UNIT_TOL, UNIT_TOL)
            for which, ln in enumerate(("message_layer_norms", "update_layer_norms")):
                so, vo = eng.debug_chain(2, layer, 2 * nt + which, s, v)
                ro, rv = O.gvp_layernorm(sd, p + f"{ln}.{name}.", s, v)
                close(so, ro, UNIT_TOL, UNIT_TOL); close(vo, rv, UNIT_TOL, UNIT_TOL)
    s, v = torch.randn(9, 128, generator=gen), torch.randn(9, 16, 3, generator=gen)
    eh, ex = eng.debug_chain(3, 0, 0, s, v)
    rh, rx = O.noise_head(sd, "dynamics.noise_predictor.noise_predictor.", cfg, s, v)
    close(eh, rh, UNIT_TOL, UNIT_TOL); close(ex, rx, UNIT_TOL, UNIT_TOL)


def test_pp_edges_as_the_reference_dataset_code_emits_them():
    """pf_build_pp_edges against build_initial_complex_graph's edges (dataset/protein_pharm_dataset.py:234-236, golden
    from the reference's function): same edges in the same order for 64 / 256 / 300-atom pockets and a 2-atom one, one
    pocket at a time and as one batch of graphs; and through the product's own build_initial_complex_graph."""
    import pharmacoforge_amd as pfa
    z = load("pp_edges.npz")
    cfg = O.DynamicsConfig()
    eng = engine_for(cfg, O.make_state_dict(cfg, 0))
    xs, ns = [], z["pocket_n_prot"].tolist()
    for i, (seed, n) in enumerate(zip(z["pocket_seeds"].tolist(), ns)):
        x, h = O.synthetic_pocket(seed, n)
        xs.append(x)
        s, d = eng.build_pp_edges(x, torch.tensor([0, n]))
        assert torch.equal(s, z[f"src_{i}"].long()) and torch.equal(d, z[f"dst_{i}"].long()), (seed, n)
        g = pfa.build_initial_complex_graph(x, h, {'pp': float(z["cutoff"]), 'pf': 8, 'fp': 8, 'ff': 9},
                                            pharm_atom_positions=torch.zeros(3, 3), pharm_atom_features=torch.zeros(3, 6))
        assert torch.equal(g.pp_src, z[f"src_{i}"].long()) and torch.equal(g.pp_dst, z[f"dst_{i}"].long())
        assert g.num_nodes('prot') == n and g.num_nodes('pharm') == 3 and g.num_nodes('prot_ph') == 0
    ptr = torch.tensor([0] + list(torch.tensor(ns).cumsum(0)))
    s, d = eng.build_pp_edges(torch.cat(xs), ptr)
    ref_s = torch.cat([z[f"src_{i}"].long() + int(ptr[i]) for i in range(len(ns))])
    ref_d = torch.cat([z[f"dst_{i}"].long() + int(ptr[i]) for i in range(len(ns))])
    assert torch.equal(s, ref_s) and torch.equal(d, ref_d)


def test_reference_edge_bookkeeping_error_case():
    """message_norm 0 with kNN pf edges: the reference indexes the protein batch vector with center indices
    (dynamics_gvp.py:220) and fails when a center index is not a valid protein index; so does pf_set_pocket_batch."""
    import pharmacoforge_amd as pfa
    cfg = O.DynamicsConfig(message_norm=0, pf_k=5)
    b = O.synthetic_batch([1, 2], [3, 4], [6, 5], cfg)              # 11 centers, 7 protein atoms
    eng = engine_for(cfg, O.make_state_dict(cfg, 0))
    with pytest.raises(pfa.PfError, match="dynamics_gvp.py:220"):
        set_batch(eng, b)


def test_single_steps_vs_oracle_config2_shape():
    """One denoising step at BASELINE config-2 graph sizes (256-atom pockets, 6 centers) for a
    small batch, against the oracle with shared noise."""
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    batch = O.synthetic_batch([100, 101, 102], 256, 6, cfg)
    T = 500
    gen = torch.Generator().manual_seed(42)
    Nf = int(batch.pharm_ptr[-1])
    noise = torch.randn(4, Nf, 9, generator=gen)
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    arr = eng.coef_array(coef, reversed(range(T)))
    x0, h0 = eng.sample(arr, 3, noise)
    ox, oh = O.sample_given_receptor(sd, cfg, batch, T, 1e-5, noise, n_steps=3)
    close(x0, ox, 1e-3, 1e-3); close(h0, oh, 1e-3, 1e-3)


def test_pp_edge_builder_matches_oracle():
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    batch = O.synthetic_batch([5, 6], 200, [3, 4], cfg)
    eng = engine_for(cfg, sd)
    s, d = eng.build_pp_edges(batch.prot_x, batch.prot_ptr)
    assert torch.equal(s, batch.pp_src) and torch.equal(d, batch.pp_dst)      # same order too


def test_properties_translation_rotation_batch_independence():
    """Size-independent properties at config-2 sizes: SE(3) equivariance of the dynamics, batch
    independence (graph i unchanged by the rest of the batch), permutation of pocket atoms."""
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    batch = O.synthetic_batch([200, 201, 202, 203], 256, [6, 3, 8, 5], cfg)
    gen = torch.Generator().manual_seed(1)
    Nf = int(batch.pharm_ptr[-1])
    x_t = 3.0 * torch.randn(Nf, 3, generator=gen)
    h_t = torch.randn(Nf, 6, generator=gen)
    t = torch.tensor([0.9, 0.5, 0.1, 1.0])
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    eh, ex = eng.dynamics(x_t, h_t, t, prot_x=batch.prot_x)
    eh, ex = eh.cpu(), ex.cpu()
    # rotation + translation
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=gen))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    shift = torch.tensor([1.5, -2.0, 0.7])
    eh2, ex2 = eng.dynamics(x_t @ q.T + shift, h_t, t, prot_x=batch.prot_x @ q.T + shift)
    close(eh2, eh, 1e-3, 1e-3); close(ex2, ex @ q.T, 1e-3, 1e-3)
    # batch independence: graph 2 alone
    sub = O.PocketBatch(batch.prot_x[512:768], batch.prot_h[512:768], torch.tensor([0, 256]), torch.tensor([0, 8]),
                        *[e - 512 for e in (batch.pp_src[(batch.pp_dst >= 512) & (batch.pp_dst < 768)],
                                            batch.pp_dst[(batch.pp_dst >= 512) & (batch.pp_dst < 768)])])
    eng2 = engine_for(cfg, sd)
    set_batch(eng2, sub)
    f0, f1 = int(batch.pharm_ptr[2]), int(batch.pharm_ptr[3])
    sh, sx = eng2.dynamics(x_t[f0:f1], h_t[f0:f1], t[2:3])
    close(sh, eh[f0:f1], 1e-4, 1e-4); close(sx, ex[f0:f1], 1e-4, 1e-4)
    # the pp edges in any order (the bind sorts them by destination, stably; destination-sorted input skips the sort): the edges of
    # a destination keep their relative order under a stable shuffle by source parity, so the sums are bitwise the same
    perm = torch.cat([torch.nonzero(batch.pp_src % 2 == 0).flatten(), torch.nonzero(batch.pp_src % 2 == 1).flatten()])
    assert not bool((batch.pp_dst[perm][1:] >= batch.pp_dst[perm][:-1]).all())
    shuffled = O.PocketBatch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src[perm], batch.pp_dst[perm])
    eng3 = engine_for(cfg, sd)
    set_batch(eng3, shuffled)
    ph, px = eng3.dynamics(x_t, h_t, t, prot_x=batch.prot_x)
    close(ph.cpu(), eh, 1e-5, 1e-5); close(px.cpu(), ex, 1e-5, 1e-5)


def test_empty_and_degenerate_graphs():
    """A graph with a single pharmacophore center (no ff edges) and one with fewer protein atoms
    than pf_k."""
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    b1 = O.synthetic_batch([300], 3, 1, cfg)
    b2 = O.synthetic_batch([301], 40, 4, cfg)
    batch = O.PocketBatch(torch.cat([b1.prot_x, b2.prot_x]), torch.cat([b1.prot_h, b2.prot_h]),
                          torch.tensor([0, 3, 43]), torch.tensor([0, 1, 5]),
                          torch.cat([b1.pp_src, b2.pp_src + 3]), torch.cat([b1.pp_dst, b2.pp_dst + 3]))
    gen = torch.Generator().manual_seed(2)
    x_t = torch.randn(5, 3, generator=gen); h_t = torch.randn(5, 6, generator=gen)
    t = torch.tensor([0.3, 0.8])
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    eh, ex = eng.dynamics(x_t, h_t, t)
    oh, ox = O.dynamics_forward(sd, cfg, batch, batch.prot_x, x_t, h_t, t)
    close(eh, oh); close(ex, ox)


@pytest.mark.parametrize("name", ["dynamics_ragged.npz", "dynamics_radius.npz"])
def test_dead_work_elimination_matches_dense_computation(name, monkeypatch):
    """Receptive-field pruning + per-source precompute are exact: the same call with both disabled (every
    layer dense, like the reference) gives the same outputs up to summation order."""
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    outs = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("PFDYN_NO_PRUNE", "1")
            monkeypatch.setenv("PFDYN_NO_PRE", "1")
        eng = engine_for(cfg, sd)
        set_batch(eng, batch, z["prot_x"])
        eh, ex = eng.dynamics(z["x_t"], z["h_t"], z["t"])
        w = eng.work_detail()
        outs.append((eh.cpu(), ex.cpu(), w))
    (eh0, ex0, w0), (eh1, ex1, w1) = outs
    torch.testing.assert_close(eh0, eh1, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(ex0, ex1, rtol=2e-5, atol=2e-5)
    assert w0["flops"] == w1["flops"] and w0["executed_flops"] <= w1["executed_flops"] <= w1["flops"]
    if cfg.pf_k > 0:                      # kNN pf edges: few active atoms, so pruning must remove work
        assert w0["executed_flops"] < 0.8 * w1["executed_flops"]
    assert w1["executed_edges_per_layer"][0] == sum(w1["edges"])           # dense: every edge of layer 0


@pytest.mark.parametrize("name", ["dynamics_ragged.npz", "dynamics_radius.npz", "dynamics_knnff.npz",
                                  "dynamics_gnorm_radius.npz", "dynamics_gnorm_knn.npz"])
def test_one_wave_per_tile_kernels(name, monkeypatch):
    """The kernels used for launches with MANY tiles (k_edge_msg / k_node_update / k_noise_head: one wave per
    32 rows, dense layers, per-source precompute) are forced here on the small golden cases, so that both
    kernel families are checked against the reference goldens."""
    monkeypatch.setenv("PFDYN_RG_ROWS_MAX", "0")
    monkeypatch.setenv("PFDYN_COOP_EDGE_MAX", "0")
    monkeypatch.setenv("PFDYN_COOP2_EDGE_MAX", "0")
    monkeypatch.setenv("PFDYN_COOP_NODE_MAX", "0")
    monkeypatch.setenv("PFDYN_NO_PRUNE", "1")
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    eng = engine_for(cfg, sd)
    set_batch(eng, batch, z["prot_x"])
    eps_h, eps_x = eng.dynamics(z["x_t"], z["h_t"], z["t"])
    close(eps_h, z["eps_h"]); close(eps_x, z["eps_x"])
    li = int(z["conv_layer_index"])
    hp, vp, hf, vf = eng.conv_layer(li, z["prot_x"], z["x_t"], z["conv_in_h_prot"], z["conv_in_v_prot"],
                                    z["conv_in_h_pharm"], z["conv_in_v_pharm"])
    close(hp, z["conv_out_h_prot"]); close(vp, z["conv_out_v_prot"])
    close(hf, z["conv_out_h_pharm"]); close(vf, z["conv_out_v_pharm"])


@pytest.mark.parametrize("name", ["dynamics_ragged.npz", "dynamics_radius.npz"])
@pytest.mark.parametrize("dense", [False, True])
def test_two_workgroups_per_cu_edge_kernel(name, dense, monkeypatch):
    """k_edge_msg_coop2 (the 4-wave kernel without weight prefetch, two workgroups per CU: launches with more tiles
    than CUs) forced on the small golden cases, on the pruned and on the dense tile lists."""
    monkeypatch.setenv("PFDYN_RG_ROWS_MAX", "0")
    monkeypatch.setenv("PFDYN_COOP_EDGE_MAX", "0")
    monkeypatch.setenv("PFDYN_COOP2_EDGE_MAX", "1000000")
    if dense:
        monkeypatch.setenv("PFDYN_NO_PRUNE", "1")
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    eng = engine_for(cfg, sd)
    set_batch(eng, batch, z["prot_x"])
    eps_h, eps_x = eng.dynamics(z["x_t"], z["h_t"], z["t"])
    close(eps_h, z["eps_h"]); close(eps_x, z["eps_x"])


@pytest.mark.parametrize("name", ["dynamics_c1.npz", "dynamics_ragged.npz", "dynamics_radius.npz", "dynamics_knnff.npz",
                                  "dynamics_gnorm_radius.npz", "dynamics_gnorm_knn.npz"])
@pytest.mark.parametrize("dense", [False, True])
def test_four_waves_per_tile_kernels(name, dense, monkeypatch):
    """The 4-wave cooperative kernels (k_edge_msg_coop / k_node_update_coop / k_node_head_coop: one 32-row tile per
    workgroup; launches between the row-group and the one-wave regimes) forced on the golden cases."""
    monkeypatch.setenv("PFDYN_RG_ROWS_MAX", "0")
    if dense:
        monkeypatch.setenv("PFDYN_NO_PRUNE", "1")
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    eng = engine_for(cfg, O.make_state_dict(cfg, int(z["wseed"])))
    set_batch(eng, batch, z["prot_x"])
    eps_h, eps_x = eng.dynamics(z["x_t"], z["h_t"], z["t"])
    close(eps_h, z["eps_h"]); close(eps_x, z["eps_x"])
    li = int(z["conv_layer_index"])
    hp, vp, hf, vf = eng.conv_layer(li, z["prot_x"], z["x_t"], z["conv_in_h_prot"], z["conv_in_v_prot"],
                                    z["conv_in_h_pharm"], z["conv_in_v_pharm"])
    close(hp, z["conv_out_h_prot"]); close(vp, z["conv_out_v_prot"])
    close(hf, z["conv_out_h_pharm"]); close(vf, z["conv_out_v_pharm"])


@pytest.mark.parametrize("name", ["dynamics_c1.npz", "dynamics_ragged.npz", "dynamics_radius.npz", "dynamics_knnff.npz",
                                  "dynamics_gnorm_radius.npz", "dynamics_gnorm_knn.npz"])
@pytest.mark.parametrize("rows_per_wave", [4, 8, "4 on two waves"])
@pytest.mark.parametrize("dense", [False, True])
def test_row_group_kernels(name, rows_per_wave, dense, monkeypatch):
    """The row-group kernels (pf_rg.hip: k_rg_edge / k_rg_node, 4 or 8 rows per wave on the 4x4x1 MFMA) forced on
    the golden cases at both widths and in the two-wave form (a 4-row group on two waves, each owning 64 of the 128
    outputs), on the pruned and on the dense tile lists, fused and separate head, plus one whole conv layer with
    non-zero node vectors."""
    monkeypatch.setenv("PFDYN_RG_ROWS_MAX", "100000000")
    monkeypatch.setenv("PFDYN_RG2_ROWS_MIN", "0" if rows_per_wave == 8 else "100000000")
    monkeypatch.setenv("PFDYN_RG_SPLIT_MAX", "100000000" if rows_per_wave == "4 on two waves" else "0")
    if dense:
        monkeypatch.setenv("PFDYN_NO_PRUNE", "1")
        monkeypatch.setenv("PFDYN_NO_FUSE_HEAD", "1")
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    eng = engine_for(cfg, O.make_state_dict(cfg, int(z["wseed"])))
    set_batch(eng, batch, z["prot_x"])
    eps_h, eps_x = eng.dynamics(z["x_t"], z["h_t"], z["t"])
    close(eps_h, z["eps_h"]); close(eps_x, z["eps_x"])
    li = int(z["conv_layer_index"])
    hp, vp, hf, vf = eng.conv_layer(li, z["prot_x"], z["x_t"], z["conv_in_h_prot"], z["conv_in_v_prot"],
                                    z["conv_in_h_pharm"], z["conv_in_v_pharm"])
    close(hp, z["conv_out_h_prot"]); close(vp, z["conv_out_v_prot"])
    close(hf, z["conv_out_h_pharm"]); close(vf, z["conv_out_v_pharm"])


@pytest.mark.parametrize("name", ["dynamics_c1.npz", "dynamics_radius.npz"])
def test_separate_head_launch(name, monkeypatch):
    """By default the last layer's node update and the noise head share one launch (k_node_head_coop); with
    PFDYN_NO_FUSE_HEAD=1 the head runs as its own kernel -- same goldens."""
    monkeypatch.setenv("PFDYN_NO_FUSE_HEAD", "1")
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    eng = engine_for(cfg, O.make_state_dict(cfg, int(z["wseed"])))
    set_batch(eng, batch, z["prot_x"])
    eps_h, eps_x = eng.dynamics(z["x_t"], z["h_t"], z["t"])
    close(eps_h, z["eps_h"]); close(eps_x, z["eps_x"])


def _rand_inputs(batch, seed, pharm_nf=6, scale=3.0):
    gen = torch.Generator().manual_seed(seed)
    nf = int(batch.pharm_ptr[-1])
    return (scale * torch.randn(nf, 3, generator=gen), torch.randn(nf, pharm_nf, generator=gen),
            torch.rand(batch.batch_size, generator=gen))


def test_large_graphs_high_degree_many_centers():
    """Edge cases of the tile machinery: 600-atom pockets (kNN candidates not register-cached), a 6 A pp cutoff
    (in-degree > 33: a destination's messages span more than two tiles), 40 centers in one graph (two pharm tiles,
    ff edges over many tiles), against the oracle."""
    cfg = O.DynamicsConfig(cutoff_pp=6.0)
    sd = O.make_state_dict(cfg, 3)
    batch = O.synthetic_batch([31, 32], 600, [40, 7], cfg)
    deg = torch.bincount(batch.pp_dst, minlength=1200)
    assert int(deg.max()) > 33
    x_t, h_t, t = _rand_inputs(batch, 5, scale=5.0)
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    eh, ex = eng.dynamics(x_t, h_t, t)
    oh, ox, edges = O.dynamics_forward(sd, cfg, batch, batch.prot_x, x_t, h_t, t, return_edges=True)
    for i, et in enumerate(O.ETYPES):
        s, d = eng.get_edges(i)
        assert edge_set(s, d) == edge_set(*edges[et]), et
    close(eh, oh, 5e-4, 5e-4); close(ex, ox, 5e-4, 5e-4)


@pytest.mark.parametrize("pf_k", [5, 0])
def test_per_graph_message_norm(pf_k):
    """message_norm = 0: sum reducer divided by (edges into the node type)/(nodes of the type) + 1 per graph
    (gvp.py:504-507), counted by the true graph of each edge."""
    cfg = O.DynamicsConfig(message_norm=0, pf_k=pf_k)
    sd = O.make_state_dict(cfg, 4)
    batch = O.synthetic_batch([41, 42, 43], 64, [4, 6, 3], cfg)
    x_t, h_t, t = _rand_inputs(batch, 6)
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    eh, ex = eng.dynamics(x_t, h_t, t)
    oh, ox = O.dynamics_forward(sd, cfg, batch, batch.prot_x, x_t, h_t, t)
    close(eh, oh); close(ex, ox)


def test_endpoint_parameterisation_step():
    """sample_p_zs_given_zt with endpoint_param_coord / endpoint_param_feat (pharmacodiff.py:413-420)."""
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    batch = O.synthetic_batch([51, 52], 64, [4, 5], cfg)
    T = 50
    gen = torch.Generator().manual_seed(8)
    noise = torch.randn(3, int(batch.pharm_ptr[-1]), 9, generator=gen)
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    arr = eng.coef_array(coef, reversed(range(T)))
    x0, h0 = eng.sample(arr, 2, noise, ep_coord=True, ep_feat=True)
    ox, oh = O.sample_given_receptor(sd, cfg, batch, T, 1e-5, noise, n_steps=2, endpoint_param_coord=True,
                                     endpoint_param_feat=True)
    close(x0, ox, 1e-3, 1e-3); close(h0, oh, 1e-3, 1e-3)


@pytest.mark.parametrize("case", ["knn_pf", "knn_ff", "graph_norm", "endpoint", "near_ties", "near_ties_k16"])
def test_fused_update_and_edge_build_variants_agree(case, monkeypatch):
    """A denoising step ends with k_step_build_fast (sampler update + the next call's edge build, one atom per
    thread, three dependent global round trips) when pf edges are kNN and pockets have at most 512 atoms.  It must
    reproduce the generic bodies (PFDYN_NO_FAST_BUILD=1: k_step_build) and the separate launches of the tile-kernel
    path bit for bit: same edge sets, same orderings, same arithmetic -- ragged batch, several steps.  (The optional tail
    launch, which runs the same fast body behind the node update + head: test_gpu_n16.py::test_tail_launch_steps_equal_separate_launches.)"""
    monkeypatch.setenv("PFDYN_NO_CENTER_HOIST", "1")       # (only the merged launch leaves the center-hoist tables: its own test compares them)
    kw = dict(ff_k=3) if case == "knn_ff" else (dict(message_norm=0) if case == "graph_norm" else (dict(pf_k=16) if case == "near_ties_k16" else {}))
    cfg = O.DynamicsConfig(**kw)
    sd = O.make_state_dict(cfg, 3)
    ties = case.startswith("near_ties")
    # (near_ties_k16: 16 neighbours per center -- the selected keys fill a whole row of 16 lanes and the extra one sits in the next row --
    # and 40 / 33 centers in a graph: several rounds of the search per wave, fp reference masks beyond 32 bits)
    batch = O.synthetic_batch([61, 62, 63, 64], [300, 256, 300, 256] if ties else 300, [40, 12, 33, 6] if case == "near_ties_k16" else [3, 8, 5, 6], cfg)
    if ties:
        # The fast body's neighbour search orders truncated distance keys (pf_stepbuild.h: knn_halfwave_keys) and must notice when
        # that is not exact: 100-120 atoms of every pocket get a twin one or two ulps away along x (both directions, so the nearer
        # twin has the larger index half of the time), ~30 an exact duplicate -- the generic body orders by (d^2, index) on the full
        # bits.  Pockets of 256 and of 300 atoms: 8 and 16 candidates per lane.
        px = batch.prot_x.clone()
        for g in range(4):
            p0, n = int(batch.prot_ptr[g]), int(batch.prot_ptr[g + 1] - batch.prot_ptr[g])
            nt = 100 if n == 256 else 120
            for j in range(nt):
                x = px[p0 + j].clone()
                step = (1 + j % 2) * (1 if (j // 2) % 2 else -1)
                x[0] = (x[0:1].view(torch.int32) + step).view(torch.float32)[0]
                px[p0 + nt + 30 + j] = x
            nd = n - (2 * nt + 30)
            px[p0 + 2 * nt + 30:p0 + n] = px[p0 + nt:p0 + nt + nd]
        src, dst = O.build_pp_edges(px, batch.prot_ptr, cfg.cutoff_pp, 100)
        batch = O.PocketBatch(px, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, src, dst)
    T = 50
    gen = torch.Generator().manual_seed(9)
    noise = torch.randn(6, int(batch.pharm_ptr[-1]), 9, generator=gen)
    ep = case == "endpoint"
    res = []
    for env in ({}, {"PFDYN_NO_FAST_BUILD": "1"}, {"PFDYN_NO_ENC_FLY": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = engine_for(cfg, sd)
        for k in env:
            monkeypatch.delenv(k)
        set_batch(eng, batch)
        coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
        arr = eng.coef_array(coef, reversed(range(T)))
        x0, h0 = eng.sample(arr, 5, noise, ep_coord=ep, ep_feat=ep)
        res.append((x0.cpu(), h0.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])      # fast == generic fused
    # separate encode / build / update launches feed the same row-group kernels through h[N][128]: same values
    torch.testing.assert_close(res[0][0], res[2][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res[0][1], res[2][1], rtol=1e-5, atol=1e-5)
    ox, oh = O.sample_given_receptor(sd, cfg, batch, T, 1e-5, noise, n_steps=5, endpoint_param_coord=ep, endpoint_param_feat=ep)
    close(res[0][0], ox, 5e-3, 5e-3); close(res[0][1], oh, 5e-3, 5e-3)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_update_and_edge_build_random_shapes_fast_equals_generic(seed, monkeypatch):
    """The latency-optimised update + build (pf_stepbuild.h: key-sorted neighbour search with its exact fallback, ff edges from
    bit masks, fp edges one per pair, 16 threads per active atom, atoms dealt by pass, edge records) against the generic body over
    random shapes it must all handle: 1..20 centers per graph (graphs with a single center included), 1..16 neighbours per
    center, pockets of 20..512 atoms (one and two passes, 8 and 16 candidates per lane, pockets smaller than k), radius and kNN ff
    edges -- four steps each, bit for bit, and against the oracle."""
    import random
    monkeypatch.setenv("PFDYN_NO_CENTER_HOIST", "1")       # (the hoist's tables equal the on-the-fly encoding up to summation order: its own test)
    rng = random.Random(seed)
    for trial in range(4):
        B = rng.choice([1, 3, 5])
        n_prot = [rng.choice([20, 70, 200, 256, 257, 400, 512]) for _ in range(B)]
        n_pharm = [rng.choice([1, 2, 6, 11, 20]) for _ in range(B)]
        kw = dict(pf_k=rng.choice([1, 3, 5, 9, 16]))
        if rng.random() < 0.3:
            kw["ff_k"] = rng.choice([1, 2, 4])
        cfg = O.DynamicsConfig(**kw)
        sd = O.make_state_dict(cfg, 40 + seed)
        batch = O.synthetic_batch([900 + 10 * seed + trial * 5 + i for i in range(B)], n_prot, n_pharm, cfg)
        T, n = 50, 4
        noise = torch.randn(n + 1, int(batch.pharm_ptr[-1]), 9, generator=torch.Generator().manual_seed(100 * seed + trial))
        coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
        res = []
        for env in ({}, {"PFDYN_NO_FAST_BUILD": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            eng = engine_for(cfg, sd)
            for k in env:
                monkeypatch.delenv(k)
            set_batch(eng, batch)
            x0, h0 = eng.sample(eng.coef_array(coef, reversed(range(T))), n, noise)
            torch.cuda.synchronize()
            eng.sample_status()
            res.append((x0.cpu(), h0.cpu()))
        what = f"seed {seed} trial {trial}: atoms {n_prot}, centers {n_pharm}, {kw}"
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), what
        ox, oh = O.sample_given_receptor(sd, cfg, batch, T, 1e-5, noise, n_steps=n)
        torch.testing.assert_close(res[0][0], ox, rtol=5e-3, atol=5e-3, msg=what)
        torch.testing.assert_close(res[0][1], oh, rtol=5e-3, atol=5e-3, msg=what)


def test_kernel_family_reporting(monkeypatch):
    """pf_debug_kernel_family reports what the launch policy chose for a layer's edge messages: the row-group kernels by
    default (the n16 form for small launches; with it off, 4 rows per wave), the 32-row tile kernels when they are switched off."""
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    batch = O.synthetic_batch([71, 72], 64, [4, 5], cfg)
    x_t, h_t, t = _rand_inputs(batch, 5)
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    eng.dynamics(x_t, h_t, t)
    assert eng.kernel_family(0) == 16 and eng.kernel_family(1) == 17       # small launches: 16-row items on four waves (pf_n16.hip); the last layer's launch also updates conv layer 0's nodes
    monkeypatch.setenv("PFDYN_N16", "0")
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    eng.dynamics(x_t, h_t, t)
    assert eng.kernel_family(0) == 4 and eng.kernel_family(1) == 4
    monkeypatch.delenv("PFDYN_N16")
    monkeypatch.setenv("PFDYN_RG2_ROWS_MIN", "0")
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    eng.dynamics(x_t, h_t, t)
    assert eng.kernel_family(0) == 8
    monkeypatch.setenv("PFDYN_RG_ROWS_MAX", "0")
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    eng.dynamics(x_t, h_t, t)
    assert eng.kernel_family(0) in (32, 128)


# ---- static hoist of conv layer 0's protein-protein messages (pf_rg.hip: EdgeParams::zs) -------------------------
HOIST_VARIANTS = {
    "default": {},
    "rows4": {"PFDYN_L0_RGA": "1", "PFDYN_L0_RGP": "1"},
    "rows4_hoisted8": {"PFDYN_L0_RGA": "1", "PFDYN_L0_RGP": "2"},
    "rows8": {"PFDYN_L0_RGA": "2", "PFDYN_RG2_ROWS_MIN": "0"},
    "tile_lists": {"PFDYN_NO_COMPACT": "1"},
    "dense_layer0": {"PFDYN_NO_PRUNE": "1"},
    "dense_layer0_rows8": {"PFDYN_NO_PRUNE": "1", "PFDYN_RG2_ROWS_MIN": "0"},
}


@pytest.mark.parametrize("variant", list(HOIST_VARIANTS))
@pytest.mark.parametrize("name", list(DYN_CASES))
def test_static_hoist_vs_golden_and_full_chain(name, variant, monkeypatch):
    """Conv layer 0 with the pp messages' first GVP hoisted (trajectory constants + per-timestep type table) against
    the reference goldens, and against the same launch shapes computing the full chain (PFDYN_NO_L0_HOIST=1).  Every
    work-list form of the layer-0 edge launch is forced: mixed 4 / 8 rows per wave, tile lists, the dense layer."""
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    for k, v in HOIST_VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    eng = engine_for(cfg, sd)
    set_batch(eng, batch, z["prot_x"])
    eps_h, eps_x = eng.dynamics(z["x_t"], z["h_t"], z["t"])
    assert eng.l0_hoist() in (4, 8, 16), "the static hoist did not run"
    close(eps_h, z["eps_h"]); close(eps_x, z["eps_x"])
    monkeypatch.setenv("PFDYN_NO_L0_HOIST", "1")
    ref = engine_for(cfg, sd)
    set_batch(ref, batch, z["prot_x"])
    rh, rx = ref.dynamics(z["x_t"], z["h_t"], z["t"])
    assert ref.l0_hoist() == 0
    torch.testing.assert_close(eps_h.cpu(), rh.cpu(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(eps_x.cpu(), rx.cpu(), rtol=2e-5, atol=2e-5)


def test_static_hoist_follows_coordinates_weights_and_features():
    """The hoisted constants are recomputed when the caller passes other protein coordinates and when the weights
    change; protein features that are not element one-hots switch the hoist off."""
    name = "dynamics_ragged.npz"
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    eng = engine_for(cfg, sd)
    set_batch(eng, batch, z["prot_x"])
    eng.dynamics(z["x_t"], z["h_t"], z["t"])
    assert eng.l0_hoist() > 0
    # other coordinates (not a rigid translate) with the same static edges
    gen = torch.Generator().manual_seed(3)
    px = z["prot_x"] + 0.3 * torch.randn(z["prot_x"].shape, generator=gen)
    eh, ex = eng.dynamics(z["x_t"], z["h_t"], z["t"], prot_x=px)
    oh, ox = O.dynamics_forward(sd, cfg, batch, px, z["x_t"], z["h_t"], z["t"])
    close(eh, oh); close(ex, ox)
    # other weights
    sd2 = O.make_state_dict(cfg, int(z["wseed"]) + 1)
    eng.set_flat_params(torch.cat([sd2[n].reshape(-1) for n, _, _ in eng.param_layout()]))     # as after an optimiser step
    eh, ex = eng.dynamics(z["x_t"], z["h_t"], z["t"], prot_x=z["prot_x"])
    assert eng.l0_hoist() > 0
    oh, ox = O.dynamics_forward(sd2, cfg, batch, z["prot_x"], z["x_t"], z["h_t"], z["t"])
    close(eh, oh); close(ex, ox)
    # soft features
    soft = batch.prot_h.clone() * 0.9 + 0.01
    b2 = O.PocketBatch(batch.prot_x, soft, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst)
    set_batch(eng, b2, z["prot_x"])
    eh, ex = eng.dynamics(z["x_t"], z["h_t"], z["t"])
    assert eng.l0_hoist() == 0
    oh, ox = O.dynamics_forward(sd2, cfg, b2, z["prot_x"], z["x_t"], z["h_t"], z["t"])
    close(eh, oh); close(ex, ox)


def test_static_hoist_type_tables_announced_or_on_arrival(monkeypatch):
    """The per-timestep type tables of the static hoist: announced by pf_sample (several launches of 64 timesteps),
    computed on arrival by un-announced pf_denoise_step calls, announced by the caller (pf_prepare_timesteps) -- the
    three give bitwise the same trajectory, and the same as a handle that saw other timesteps before."""
    monkeypatch.setenv("PFDYN_NO_CENTER_HOIST", "1")       # (the center hoist also reads the announced plan and changes the summation order of the calls it serves)
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    batch = O.synthetic_batch([7, 8, 9], 96, [3, 5, 4], cfg)
    T, n = 500, 150
    Nf = int(batch.pharm_ptr[-1])
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(1))
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)

    def fresh():
        eng = engine_for(cfg, sd)
        set_batch(eng, batch)
        return eng, eng.coef_array(coef, reversed(range(n)))         # the last n steps of the schedule

    eng, arr = fresh()
    x_a, h_a = eng.sample(arr, n, noise)
    assert eng.l0_hoist() > 0
    results = []
    for announce in (False, True):
        eng, arr = fresh()
        if announce:
            eng.prepare_timesteps(arr, n)
        else:                                                        # other timesteps first: the cache is not empty
            eng.prepare_timesteps(eng.coef_array(coef, range(200, 330)), 130)
        eng.sample_begin(noise[0])
        for i in range(n):
            eng.denoise_step(arr[i], noise[i + 1])
        results.append(eng.sample_frame())
    for x, h in results:
        assert torch.equal(x.cpu(), x_a.cpu()) and torch.equal(h.cpu(), h_a.cpu())
