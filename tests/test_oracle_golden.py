"""The CPU oracle (oracle/pf_oracle.py) against golden vectors recorded from the reference's own
code (tests/golden/make_golden.py).  Tolerances: fp32, same op order up to BLAS summation order."""
import numpy as np
import pytest
import torch

from oracle import pf_oracle as O
from helpers import DYN_CASES, GRAD_CASES, batch_from, dropout_from, edge_set, load

RTOL, ATOL = 1e-5, 1e-5


def close(a, b, rtol=RTOL, atol=ATOL):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def units():
    return load("units.npz"), O.make_state_dict(O.DynamicsConfig(), 0)


def test_gvp_shapes(units):
    z, sd = units
    p = "dynamics.noise_predictor.conv_layers.0."
    so, vo = O.gvp_forward(sd, p + "edge_message_fns.prot_pp_prot.0.", z["msg0_s"], z["msg0_v"])
    close(so, z["msg0_so"]); close(vo, z["msg0_vo"])
    assert torch.isfinite(vo).all()
    so, vo = O.gvp_forward(sd, p + "edge_message_fns.pharm_ff_pharm.1.", z["msg1_s"], z["msg1_v"])
    close(so, z["msg1_so"]); close(vo, z["msg1_vo"])
    so, vo = O.gvp_chain(sd, p + "edge_message_fns.prot_pf_pharm.", 3, z["chain_s"], z["chain_v"])
    close(so, z["chain_so"]); close(vo, z["chain_vo"])
    so, vo = O.gvp_chain(sd, p + "node_update_fns.prot.", 2, z["upd_s"], z["upd_v"])
    close(so, z["upd_so"]); close(vo, z["upd_vo"])


def test_layernorm_rbf_head_encoders(units):
    z, sd = units
    p = "dynamics.noise_predictor.conv_layers.0."
    so, vo = O.gvp_layernorm(sd, p + "message_layer_norms.pharm.", z["ln_s"], z["ln_v"])
    close(so, z["ln_so"]); close(vo, z["ln_vo"])
    assert torch.all(vo[5] == 0)
    close(O.rbf(z["rbf_d"], D_max=15.0, D_count=16), z["rbf_out"], atol=1e-7)
    eh, ex = O.noise_head(sd, "dynamics.noise_predictor.noise_predictor.", O.DynamicsConfig(), z["head_s"], z["head_v"])
    close(eh, z["head_eh"]); close(ex, z["head_ex"])
    for nt in ("pharm", "prot"):
        x = z[f"enc_{nt}_in"]
        close(O.encode(sd, f"dynamics.{nt}_encoder.", x[:, :-1], x[:, -1]), z[f"enc_{nt}_out"])


def test_state_dict_spec_matches_reference_layout():
    # 244 tensors under dynamics.* and 772,815 parameters at dev.yml
    spec = O.state_dict_spec(O.DynamicsConfig())
    assert len(spec) == 244
    assert sum(int(np.prod(s)) for _, s, _ in spec) == 772815
    spec = O.state_dict_spec(O.DynamicsConfig(n_convs=4, n_noise_gvps=3))
    assert sum(int(np.prod(s)) for _, s, _ in spec) == 1447879


@pytest.mark.parametrize("T", [50, 100, 500, 1000])
@pytest.mark.parametrize("prec", [1e-5, 1e-4])
def test_schedule_tables(T, prec):
    z = load("schedule.npz")
    tag = f"T{T}_p{prec:g}"
    gamma = O.gamma_table(T, prec)
    assert torch.equal(gamma, z["gamma_" + tag])          # float64 numpy pipeline -> bit exact
    c = O.step_coefficients(gamma, T)
    # fp32 torch expm1 / softplus / exp: 1 ulp between the vectorised CPU kernels of different instruction sets
    for k, zk in (("alpha_t_given_s", "a_ts_"), ("var_terms", "var_"), ("sigma", "sigma_")):
        torch.testing.assert_close(c[k], z[zk + tag], rtol=1e-6, atol=1e-9)


def test_schedule_probe_values():
    # SURVEY.md 8(c)(3) probe values
    g = O.gamma_table(500, 1e-5)
    assert abs(float(g[0]) + 11.5129) < 1e-3 and abs(float(g[500]) - 11.511) < 1e-3
    c = O.step_coefficients(g, 500)
    np.testing.assert_allclose([1 / float(c["alpha_t_given_s"][499]), float(c["var_terms"][499]), float(c["sigma"][499])],
                               [1.610165, 0.989116, 0.783761], rtol=2e-5)
    np.testing.assert_allclose([1 / float(c["alpha_t_given_s"][0]), float(c["var_terms"][0]), float(c["sigma"][0])],
                               [1.000004, 0.001886, 0.002108], rtol=5e-4)


@pytest.mark.parametrize("name", list(DYN_CASES))
def test_edges_conv_dynamics(name):
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    # edge sets produced by the reference (torch_cluster stand-in written independently)
    edges = O.build_dynamic_edges(cfg, batch, z["prot_x"], z["x_t"])
    edges["pp"] = (batch.pp_src, batch.pp_dst)
    for et in O.ETYPES:
        assert edge_set(*edges[et]) == edge_set(z[f"e_{et}_src"].long(), z[f"e_{et}_dst"].long()), et
        assert len(edges[et][0]) == len(z[f"e_{et}_src"])
    # one conv layer with non-zero vectors, on the reference's edge lists
    li = int(z["conv_layer_index"])
    ref_edges = {et: (z[f"e_{et}_src"].long(), z[f"e_{et}_dst"].long()) for et in O.ETYPES}
    nf = {"pharm": (z["conv_in_h_pharm"], z["x_t"], z["conv_in_v_pharm"]),
          "prot": (z["conv_in_h_prot"], z["prot_x"], z["conv_in_v_prot"])}
    ec = O.dynamic_edge_counts(cfg, batch, ref_edges) if cfg.message_norm == 0 else None
    out = O.conv_layer(sd, f"dynamics.noise_predictor.conv_layers.{li}.", cfg, nf, ref_edges, batch, ec)
    for nt in ("pharm", "prot"):
        close(out[nt][0], z[f"conv_out_h_{nt}"], rtol=1e-4, atol=2e-5)
        close(out[nt][2], z[f"conv_out_v_{nt}"], rtol=1e-4, atol=2e-5)
    # the boundary function
    eps_h, eps_x = O.dynamics_forward(sd, cfg, batch, z["prot_x"], z["x_t"], z["h_t"], z["t"])
    close(eps_h, z["eps_h"], rtol=1e-4, atol=2e-5)
    close(eps_x, z["eps_x"], rtol=1e-4, atol=2e-5)


def test_reference_edge_bookkeeping_for_per_graph_norm():
    """message_norm = 0 reads the per-graph edge counts add_pharm_edges stored (dynamics_gvp.py:218-225).  With kNN pf
    edges the reference books center j's edges on the graph that owns PROTEIN atom j (:220); the golden was produced
    by that code, so the oracle must reproduce the same counts -- and they differ from the true per-graph counts."""
    z, cfg = load("dynamics_gnorm_knn.npz"), DYN_CASES["dynamics_gnorm_knn.npz"]
    batch = batch_from(z)
    edges = {et: (z[f"e_{et}_src"].long(), z[f"e_{et}_dst"].long()) for et in O.ETYPES}
    ec = O.dynamic_edge_counts(cfg, batch, edges)
    # pockets of 5 / 40 / 30 atoms, 4 / 6 / 3 centers, k = 5: centers 0..4 -> graph 0, centers 5..12 -> graph 1
    assert ec["pf"].tolist() == [25, 40, 0] and ec["fp"].tolist() == [25, 40, 0]
    true_pf = torch.bincount(torch.searchsorted(batch.pharm_ptr[1:].contiguous(), edges["pf"][1], right=True), minlength=3)
    assert true_pf.tolist() == [20, 30, 15]
    assert ec["ff"].tolist() == torch.bincount(torch.searchsorted(batch.pharm_ptr[1:].contiguous(), edges["ff"][1], right=True),
                                               minlength=3).tolist()
    # radius pf edges: the lookup uses protein indices, the counts are the true ones
    z, cfg = load("dynamics_gnorm_radius.npz"), DYN_CASES["dynamics_gnorm_radius.npz"]
    batch = batch_from(z)
    edges = {et: (z[f"e_{et}_src"].long(), z[f"e_{et}_dst"].long()) for et in O.ETYPES}
    ec = O.dynamic_edge_counts(cfg, batch, edges)
    true_pf = torch.bincount(torch.searchsorted(batch.pharm_ptr[1:].contiguous(), edges["pf"][1], right=True), minlength=3)
    assert ec["pf"].tolist() == true_pf.tolist() and sum(ec["pf"].tolist()) == edges["pf"][0].numel()


@pytest.mark.parametrize("name,traj,ep", [("traj_c1.npz", True, False), ("traj_ragged.npz", False, False),
                                          ("traj_c1_T500.npz", True, False), ("traj_endpoint.npz", True, True),
                                          ("traj_c1_T500_bounded.npz", True, False)])
def test_trajectory(name, traj, ep):
    """sample_given_receptor against the reference's trajectories: config 1 at T=50 and over the whole T=500 schedule
    (every frame; once at the shipped schedule precision, where random weights let the centers drift ~480 A away, once
    at precision 0.25, where they stay inside the pocket at every step), a ragged batch, and the endpoint
    parameterisation (pharmacodiff.py:413-420)."""
    z = load(name)
    cfg = O.DynamicsConfig()
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    T = int(z["T"])
    prec = float(z["precision"]) if "precision" in z else 1e-5
    res = O.sample_given_receptor(sd, cfg, batch, T, prec, z["noise"], return_traj=traj, endpoint_param_coord=ep,
                                  endpoint_param_feat=ep)
    # T-step stochastic trajectory: rounding differences compound, tolerance is looser
    close(res[0], z["x0"], rtol=1e-3, atol=1e-3)
    close(res[1], z["h0"], rtol=1e-3, atol=1e-3)
    if traj:
        pos = torch.stack([f[0] for f in res[2]])
        feat = torch.stack([f[1] for f in res[2]])
        close(pos, z["pos_frames"], rtol=1e-3, atol=1e-3)
        close(feat, z["feat_frames"], rtol=1e-3, atol=1e-3)


def test_sample_multi_pocket_and_copy_graph():
    """PharmacophoreDiff.sample (pharmacodiff.py:516-578) + copy_graph (utils/unorganized_utils.py:28-81): three pockets
    of different sizes, ragged requests, batches of 4 in list order, explicit init_pharm_com."""
    z = load("sample_multi.npz")
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    pockets = [O.synthetic_batch([int(s)], int(n), 1, cfg) for s, n in zip(z["pocket_seeds"], z["pocket_n_prot"])]
    sizes, per = z["n_pharms_flat"].tolist(), z["n_pharms_per_pocket"].tolist()
    n_pharms, k = [], 0
    for c in per:
        n_pharms.append(sizes[k:k + c]); k += c
    noises = [z[f"noise_{i}"] for i in range(2)]
    out = O.sample(sd, cfg, pockets, n_pharms, int(z["max_batch_size"]), int(z["T"]), 1e-5, noises, z["init_pharm_com"])
    assert [len(o) for o in out] == per
    x0 = torch.cat([x for o in out for x, _ in o]); h0 = torch.cat([h for o in out for _, h in o])
    close(x0, z["x0"], rtol=1e-3, atol=1e-3); close(h0, z["h0"], rtol=1e-3, atol=1e-3)


def test_static_pp_edges_as_build_initial_complex_graph_emits_them():
    """dataset/protein_pharm_dataset.py:234-236 on single pockets (64 / 256 / 300 atoms and a 2-atom corner case):
    the oracle's radius graph gives the same edges in the same order."""
    z = load("pp_edges.npz")
    for i, (seed, n) in enumerate(zip(z["pocket_seeds"].tolist(), z["pocket_n_prot"].tolist())):
        x, _ = O.synthetic_pocket(seed, n)
        s, d = O.build_pp_edges(x, torch.tensor([0, n]), float(z["cutoff"]), 100)
        assert torch.equal(s, z[f"src_{i}"].long()) and torch.equal(d, z[f"dst_{i}"].long()), (seed, n)


def test_training_forward():
    z = load("train_fwd.npz")
    cfg = O.DynamicsConfig()
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    losses, metrics = O.training_forward(sd, cfg, batch, z["x0"], z["h0"], int(z["T"]), 1e-5,
                                         z["t_int"].long(), z["eps_h"], z["eps_x"])
    for k, v in {**losses, **metrics}.items():
        ref = float(z["out_" + k.replace(" ", "_")])
        assert abs(float(v) - ref) <= 1e-4 * max(1.0, abs(ref)), (k, float(v), ref)


@pytest.mark.parametrize("name", sorted(GRAD_CASES))
def test_training_gradients(name):
    """Loss and every parameter gradient of one reference training_step (train() mode, dropout 0.1)."""
    z = load(name)
    cfg = GRAD_CASES[name]
    batch = batch_from(z)
    sd = O.make_state_dict(cfg, int(z["wseed"]))
    losses, metrics, grads = O.training_grads(sd, cfg, batch, z["x0"], z["h0"], int(z["T"]), 1e-5,
                                              z["t_int"].long(), z["eps_h"], z["eps_x"],
                                              dropout=dropout_from(z, cfg), weighted_loss=bool(z["weighted_loss"]))
    for k, v in {**losses, **metrics}.items():
        ref = float(z["out_" + k.replace(" ", "_")])
        assert abs(float(v) - ref) <= 1e-5 * max(1.0, abs(ref)), (k, float(v), ref)
    live = 0
    for k, g in grads.items():
        if g.numel() == 0:
            continue
        ref = z["grad_" + k]
        scale = float(ref.abs().max())
        live += scale > 0
        assert float((g - ref).abs().max()) <= 2e-5 * scale + 1e-9, k
    assert live >= 150          # the last layer's prot-side parameters get exactly zero gradient


def test_neighbour_edge_cases():
    # empty graphs, single nodes, k larger than the candidate set, exact-radius exclusion
    x = torch.tensor([[0., 0, 0], [3.5, 0, 0], [0, 1.0, 0]])
    e = O.radius_graph(x, 3.5, torch.tensor([0, 3]), 100)
    assert edge_set(e[0], e[1]) == {(2, 0), (0, 2)}            # d == r is excluded (strict <)
    assert O.radius_graph(x[:1], 3.5, torch.tensor([0, 1]), 100).shape == (2, 0)
    k = O.knn(x, torch.zeros(1, 3), 5, torch.tensor([0, 3]), torch.tensor([0, 1]))
    assert k[1].tolist() == [0, 2, 1]                           # only 3 candidates, ascending distance
    k = O.knn(x, torch.zeros(2, 3), 1, torch.tensor([0, 0, 3]), torch.tensor([0, 1, 2]))
    assert k.tolist() == [[1], [0]]                             # first graph has no candidates
    e = O.radius_graph(torch.zeros(4, 3), 1.0, torch.tensor([0, 4]), 2)
    assert e.shape[1] == 8                                      # max_num_neighbors truncates per target
