"""GPU parity tests of the n16 kernels (pharmacophore-diffusion_amd/csrc/pf_n16.hip: 16-row items on the four waves of a
workgroup, v_mfma_f32_16x16x4_f32), through the C ABI: the chain code in isolation against the reference's own module
outputs (tests/golden/units.npz) and the oracle, and the kernels forced onto every dynamics golden.

Tolerances as in test_gpu_parity.py (fp32 both sides): a chain 5e-5, a dynamics call 2e-4 + 2e-4 |ref|."""
import pytest
import torch

from oracle import pf_oracle as O
from helpers import DYN_CASES, batch_from, load
from test_gpu_parity import UNIT_TOL, close, engine_for, set_batch

pytestmark = pytest.mark.gpu

ET_NAMES = ["pharm_ff_pharm", "prot_pf_pharm", "pharm_fp_prot", "prot_pp_prot"]


def test_n16_chain_units_vs_reference_modules():
    """pf_debug_chain kinds 16 / 17 (n16_block on supplied rows) against the reference's own GVP chains
    (gvp.py:89-116; rows of tests/golden/units.npz): the 3-GVP message chain with a 17-channel first GVP and the 2-GVP
    update chain, 37 rows = two full items and a ragged one."""
    z = load("units.npz")
    cfg = O.DynamicsConfig()
    eng = engine_for(cfg, O.make_state_dict(cfg, 0))
    so, vo = eng.debug_chain(16, 0, 1, z["chain_s"], z["chain_v"])           # edge_message_fns.prot_pf_pharm
    close(so, z["chain_so"], UNIT_TOL, UNIT_TOL); close(vo, z["chain_vo"], UNIT_TOL, UNIT_TOL)
    so, vo = eng.debug_chain(17, 0, 0, z["upd_s"], z["upd_v"])               # node_update_fns.prot
    close(so, z["upd_so"], UNIT_TOL, UNIT_TOL); close(vo, z["upd_vo"], UNIT_TOL, UNIT_TOL)


@pytest.mark.parametrize("arch", ["dev", "deep"])
def test_n16_every_chain_vs_oracle(arch):
    """Every message chain (layer x edge type) and update chain (layer x node type) in the n16 form against the oracle on
    random rows; row counts 1..40 cover every remainder of the 16-row item."""
    cfg = O.DynamicsConfig() if arch == "dev" else O.DynamicsConfig(n_convs=3, n_message_gvps=2, n_update_gvps=3, n_noise_gvps=2)
    sd = O.make_state_dict(cfg, 7)
    eng = engine_for(cfg, sd)
    gen = torch.Generator().manual_seed(3)
    counts = iter([1, 2, 3, 5, 7, 8, 9, 13, 15, 16, 17, 21, 27, 31, 32, 33, 40, 4, 6, 10, 11, 12, 14, 18] * 2)
    for layer in range(cfg.n_convs):
        p = f"dynamics.noise_predictor.conv_layers.{layer}."
        for et in range(4):
            n = next(counts)
            s, v = torch.randn(n, 144, generator=gen), torch.randn(n, 17, 3, generator=gen)
            s[:, 128:] = s[:, 128:].abs().clamp(max=1.0)
            so, vo = eng.debug_chain(16, layer, et, s, v)
            ro, rv = O.gvp_chain(sd, p + f"edge_message_fns.{ET_NAMES[et]}.", cfg.n_message_gvps, s, v)
            close(so, ro, UNIT_TOL, UNIT_TOL); close(vo, rv, UNIT_TOL, UNIT_TOL)
        for nt, name in enumerate(("prot", "pharm")):
            n = next(counts)
            s, v = torch.randn(n, 128, generator=gen), torch.randn(n, 16, 3, generator=gen)
            so, vo = eng.debug_chain(17, layer, nt, s, v)
            ro, rv = O.gvp_chain(sd, p + f"node_update_fns.{name}.", cfg.n_update_gvps, s, v)
            close(so, ro, UNIT_TOL, UNIT_TOL); close(vo, rv, UNIT_TOL, UNIT_TOL)


@pytest.mark.parametrize("mask", [1, 2, 3, 5, 7])
@pytest.mark.parametrize("variant", ["compact", "tile_lists", "dense"])
@pytest.mark.parametrize("name", list(DYN_CASES))
def test_n16_edge_kernels_on_goldens(name, variant, mask, monkeypatch):
    """The n16 edge kernel forced onto every dynamics golden (PFDYN_N16 bit 0: conv layers >= 1, first message GVP reads
    h / v from memory; bit 1: conv layer 0, protein sources from the static hoist's type tables, centers encoded on the
    fly; bit 2: with two conv layers, conv layer 0's node update fused into the last layer's edge launch), on compact work lists, on tile lists and on the dense (unpruned) lists; the node kernels read its 16-slot
    partial rows."""
    monkeypatch.setenv("PFDYN_N16", str(mask))
    monkeypatch.setenv("PFDYN_N16_ROWS_MAX", "100000000")
    if variant == "tile_lists":
        monkeypatch.setenv("PFDYN_NO_COMPACT", "1")
    if variant == "dense":
        monkeypatch.setenv("PFDYN_NO_PRUNE", "1")
    z, cfg = load(name), DYN_CASES[name]
    batch = batch_from(z)
    eng = engine_for(cfg, O.make_state_dict(cfg, int(z["wseed"])))
    set_batch(eng, batch, z["prot_x"])
    eps_h, eps_x = eng.dynamics(z["x_t"], z["h_t"], z["t"])
    fusable = cfg.n_convs == 2 and cfg.pf_k > 0 and variant == "compact"      # bit 2: conv layer 0's node update inside the last layer's edge launch
    if mask & 1:
        assert eng.kernel_family(cfg.n_convs - 1) == (17 if (mask & 4) and fusable else 16)
    if mask & 2:
        assert eng.kernel_family(0) == 16 and eng.l0_hoist() == 16
    close(eps_h, z["eps_h"]); close(eps_x, z["eps_x"])


def test_n16_with_pocket_sharing(monkeypatch):
    """Copies of a pocket share conv layer 0's protein -> protein messages under the n16 edge kernel too: a
    30-step trajectory of 2 pockets x (9, 7) ragged copies, shared == per-copy == the row-group kernels, and the shared
    launch computes fewer edges."""
    from test_gpu_fullsize import _copies_batch, _engine
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 0)
    sizes = [3, 8, 5, 4, 6, 7, 3, 5, 4]
    batch, uid = _copies_batch(cfg, [(511, 120), (512, 90)], [sizes, sizes[:7]])
    T, n = 100, 30
    Nf = int(batch.pharm_ptr[-1])
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(5))
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    res, fam, work = {}, {}, {}
    for name, env, shared in (("rg", {"PFDYN_N16": "0"}, True), ("n16", {"PFDYN_N16": "7"}, False), ("n16_shared", {"PFDYN_N16": "7"}, True)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = _engine(cfg, sd)
        eng.set_batch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst,
                      pocket_uid=uid if shared else None)
        x, h = eng.sample(eng.coef_array(coef, reversed(range(n))), n, noise)
        res[name], fam[name], work[name] = (x.cpu(), h.cpu()), eng.kernel_family(0), eng.work_detail()
    assert fam == {"rg": 4, "n16": 16, "n16_shared": 16}, fam          # (layer 0; the last layer runs the fused launch in both n16 cases)
    for a in ("rg", "n16"):
        torch.testing.assert_close(res["n16_shared"][0], res[a][0], rtol=2e-4, atol=2e-4)
        torch.testing.assert_close(res["n16_shared"][1], res[a][1], rtol=2e-4, atol=2e-4)
    assert work["n16_shared"]["executed_edges_per_layer"][0] < 0.9 * work["n16"]["executed_edges_per_layer"][0]


@pytest.mark.parametrize("norm", ["mean", 0, 4.0])
def test_n16_default_policy_on_random_ragged_batches_vs_oracle(norm):
    """The default launch policy (n16 edge kernels, fused launch) on seeded random ragged batches against the oracle: 1-12
    graphs, pockets of 3-70 atoms (some smaller than k), 1-10 centers (a single center has no ff edges: its graph's store
    item is the only writer of that center's conv-layer-0 update), both calling conventions (one common t, per-graph t),
    centers placed inside and outside the ff cutoff."""
    import random
    rnd = random.Random(11)
    cfg = O.DynamicsConfig(message_norm=norm)
    sd = O.make_state_dict(cfg, 5)
    eng = engine_for(cfg, sd)
    for trial in range(6):
        B = rnd.choice([1, 2, 3, 5, 8, 12])
        n_prot = [rnd.choice([3, 4, 7, 12, 20, 33, 48, 70]) for _ in range(B)]
        n_pharm = [rnd.choice([1, 1, 2, 3, 5, 8, 10]) for _ in range(B)]
        batch = O.synthetic_batch([900 + 20 * trial + i for i in range(B)], n_prot, n_pharm, cfg)
        gen = torch.Generator().manual_seed(100 + trial)
        Nf = int(batch.pharm_ptr[-1])
        spread = rnd.choice([1.0, 4.0, 12.0])                           # 12 A: most center pairs beyond the 9 A ff cutoff
        x_t, h_t = spread * torch.randn(Nf, 3, generator=gen), torch.randn(Nf, 6, generator=gen)
        set_batch(eng, batch)
        for t in (torch.full((B,), 0.37), torch.rand(B, generator=gen)):
            eps_h, eps_x = eng.dynamics(x_t, h_t, t)
            assert eng.kernel_family(0) == 16 and eng.kernel_family(1) == 17, (trial, eng.kernel_family(0), eng.kernel_family(1))
            oh, ox = O.dynamics_forward(sd, cfg, batch, batch.prot_x, x_t, h_t, t)
            close(eps_h, oh); close(eps_x, ox)


@pytest.mark.parametrize("case", ["config1", "ragged", "knnff", "gnorm", "deep", "many_centers"])
def test_tail_launch_steps_equal_separate_launches(case, monkeypatch):
    """The tail launch -- the centers' node update of the last conv layer, the noise head, the p(z_s | z_t) update and the next
    call's edge build in ONE launch, one workgroup per graph (pharmacodiff.py:395-429, dynamics_gvp.py:37-42,187-227) -- in
    both forms (pf_rg.hip: k_rg_tail, two two-wave items of four centers, the default; pf_n16.hip: k_n16_tail, a 16-row item
    with to_scalar_output packed into the padded last head GVP) against the separate node + head and update + build launches
    on the same states: three denoising steps each, state (x_t, h_t) within the single-call
    tolerance after every step (steps 2 and 3 consume the edges the previous step's tail built) and the dynamic edge lists
    of the next call IDENTICAL (same order: both forms run the same build body).  Cases: config-1 shape, ragged pockets
    (some smaller than k) with 1-10 centers, kNN ff edges, per-graph normalisers, 3 conv layers / 2 head GVPs, and graphs
    with more than 16 centers (two passes of the 16-row item)."""
    cfg = {"config1": O.DynamicsConfig(), "ragged": O.DynamicsConfig(), "knnff": O.DynamicsConfig(ff_k=2, pf_k=3, message_norm=1),
           "gnorm": O.DynamicsConfig(message_norm=0, pf_k=5),
           "deep": O.DynamicsConfig(n_convs=3, n_message_gvps=2, n_update_gvps=3, n_noise_gvps=2),
           "many_centers": O.DynamicsConfig()}[case]
    sd = O.make_state_dict(cfg, 3)
    if case == "config1":
        n_prot, n_pharm = [64], [4]
    elif case == "many_centers":
        n_prot, n_pharm = [40, 90, 33], [17, 33, 5]
    else:
        n_prot, n_pharm = [3, 70, 12, 48, 7, 33], [1, 10, 3, 8, 2, 5]
    batch = O.synthetic_batch([700 + i for i in range(len(n_prot))], n_prot, n_pharm, cfg)
    Nf = int(batch.pharm_ptr[-1])
    T, n = 100, 3
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(17))
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    states, edges = {}, {}
    # ("merged": the default -- the node + head items and every graph's update + build as workgroups of ONE launch, k_rg_node_hs_build,
    # the noise prediction handed over through polled exchange words; it applies where the static tiling of the centers does)
    for form, mask, tform, fam, hsb in (("tail_rg", "15", "rg", 4, "0"), ("tail_n16", "15", "n16", 16, "0"), ("separate", "7", "rg", 0, "0"),
                                        ("merged", "7", "rg", None, "1")):
        monkeypatch.setenv("PFDYN_N16", mask)
        monkeypatch.setenv("PFDYN_TAIL_FORM", tform)
        monkeypatch.setenv("PFDYN_HS_BUILD", hsb)
        eng = engine_for(cfg, sd)
        set_batch(eng, batch)
        arr = eng.coef_array(coef, [40, 39, 38])
        eng.sample_begin(noise[0])
        st, ed = [], []
        for i in range(n):
            eng.denoise_step(arr[i], noise[i + 1])
            if fam is not None:
                assert eng.kernel_family(cfg.n_convs) == fam
            else:
                merged_ran = eng.kernel_family(cfg.n_convs) == 2
                assert eng.kernel_family(cfg.n_convs) in (0, 2)
            x, h = eng.sample_frame()
            st.append((x.cpu(), h.cpu()))
            ed.append([tuple(t.clone() for t in eng.get_edges(et)) for et in range(3)])
        states[form], edges[form] = st, ed
        assert eng.xchg_timeouts() == 0
    if case in ("config1", "ragged", "gnorm", "many_centers"):
        assert merged_ran, case                        # two conv layers, kNN pf edges, a small batch: the merged launch is the default
    for i in range(n):
        # the merged launch runs the code of the separate launches (the same items, the same build body, eps handed over bit for
        # bit) compiled into another kernel: the compiler may contract a multiply-add differently there, so equal to an ulp or two
        for a, b in zip(states["merged"][i], states["separate"][i]):
            torch.testing.assert_close(a, b, rtol=2e-6 * (i + 1), atol=2e-6 * (i + 1))
        for et in range(3):
            (s1, d1), (s2, d2) = edges["merged"][i][et], edges["separate"][i][et]
            assert torch.equal(s1, s2) and torch.equal(d1, d2), (case, "merged", i, et)
        for form in ("tail_rg", "tail_n16"):
            for a, b in zip(states[form][i], states["separate"][i]):
                torch.testing.assert_close(a, b, rtol=2e-4 * (i + 1), atol=2e-4 * (i + 1))
            for et in range(3):
                (s1, d1), (s2, d2) = edges[form][i][et], edges["separate"][i][et]
                assert torch.equal(s1, s2) and torch.equal(d1, d2), (case, form, i, et)
        # the row-group tail runs the very code of the separate node + head launch: same bits
        for a, b in zip(states["tail_rg"][i], states["separate"][i]):
            assert torch.equal(a, b), (case, i)


@pytest.mark.parametrize("case", ["config1", "ragged", "gnorm", "uniform"])
def test_center_hoist_equals_on_the_fly_encoding(case, monkeypatch):
    """Center hoist (pf_cenhoist.h; CenHoistParams): with the next call's timestep announced (pf_prepare_timesteps; pf_sample does it),
    a denoising step's merged last launch also leaves every center's encoder output h_c and P_et[c] = W_et[:, :128] h_c + b for the
    etypes whose source is a center (ff, fp; dynamics_gvp.py:107-117, gvp.py:545-549) -- computed by workgroups of their own from the
    head's eps_h and a snapshot of the features, under the update + build -- and the next call's conv-layer-0 items of those etypes
    start from a row of P (n16 kind M0H) instead of encoding their sources and running the full first message GVP (M0Z); the fused
    launch reads h_c as the centers' residual input.  Same values up to fp32 summation order: eight steps with and without the
    tables agree at the tolerance of a call per step, and with the oracle; the work-list and the arithmetic item maps both."""
    cfg = {"config1": O.DynamicsConfig(), "ragged": O.DynamicsConfig(), "gnorm": O.DynamicsConfig(message_norm=0, pf_k=5),
           "uniform": O.DynamicsConfig()}[case]
    sd = O.make_state_dict(cfg, 9)
    n_prot, n_pharm = {"config1": ([64], [4]), "uniform": ([96] * 5, [6] * 5)}.get(case, ([30, 70, 12, 48, 7, 33], [1, 10, 3, 8, 2, 5]))
    batch = O.synthetic_batch([800 + i for i in range(len(n_prot))], n_prot, n_pharm, cfg)
    Nf = int(batch.pharm_ptr[-1])
    T, n = 100, 8
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(23))
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    res = {}
    for form in ("hoist", "encode"):
        if form == "encode":
            monkeypatch.setenv("PFDYN_NO_CENTER_HOIST", "1")
        eng = engine_for(cfg, sd)
        set_batch(eng, batch)
        x, h = eng.sample(eng.coef_array(coef, reversed(range(n))), n, noise)          # s = 7 ... 0, announced as a plan
        torch.cuda.synchronize()
        eng.sample_status()
        assert eng.kernel_family(0) == 16 and eng.kernel_family(cfg.n_convs + 1) == (1 if form == "hoist" else 0)
        assert eng.xchg_timeouts() == 0
        res[form] = (x.cpu(), h.cpu())
    for a, b in zip(res["hoist"], res["encode"]):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-3)
    # the oracle on the same steps (s = n - 1 ... 0: the end of the schedule)
    bidx = batch.batch_idxs()
    init_com = O.segment_mean(batch.prot_x, batch.prot_ptr)
    px = batch.prot_x - init_com[bidx["prot"]]
    x_t, h_t = noise[0][:, :3].clone(), noise[0][:, 3:].clone()
    with torch.no_grad():
        for i, s in enumerate(reversed(range(n))):
            px, x_t, h_t = O.sample_step(sd, cfg, batch, coef, s, px, x_t, h_t, noise[1 + i][:, :3], noise[1 + i][:, 3:])
    ref_x = x_t - O.segment_mean(px, batch.prot_ptr)[bidx["pharm"]] + init_com[bidx["pharm"]]
    torch.testing.assert_close(res["hoist"][0], ref_x, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(res["hoist"][1], h_t, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("case", ["uniform", "ragged"])
def test_pa_messages_computed_ahead_equal_the_launch_s_own(case, monkeypatch):
    """Speculative "pa" messages (k_n16_pa_spec; BuildParams::pa_same): conv layer 0's messages along the pp edges into the active
    atoms depend on the timestep, the element types and the static pocket geometry only, so a denoising step computes the NEXT call's
    on a side stream under its own last launch, and the next call skips the "pa" region of every graph whose region came out unchanged
    (same atoms, same slots: stamped per atom by the update + build).  Same item code, same tables, same slots: a run with the
    speculation equals a run without it (PFDYN_NO_PA_SPEC=1) BIT FOR BIT -- 40 steps from the noisy end of the schedule (regions change
    every step: nothing may be skipped wrongly) and 40 from its quiet end (most regions stay: rows computed ahead are what the fused
    launch reads), trajectories included; and the quiet run did skip."""
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 5)
    n_prot, n_pharm = ([96] * 6, [6] * 6) if case == "uniform" else ([30, 70, 112, 48, 64, 33], [3, 8, 5, 6, 4, 7])
    batch = O.synthetic_batch([820 + i for i in range(len(n_prot))], n_prot, n_pharm, cfg)
    Nf = int(batch.pharm_ptr[-1])
    T, n = 500, 40
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(29))
    coef = O.step_coefficients(O.gamma_table(T, 0.25), T)           # (bounded schedule: the centers stay inside the pocket)
    res, skipped = {}, {}
    for form in ("spec", "plain"):
        if form == "plain":
            monkeypatch.setenv("PFDYN_NO_PA_SPEC", "1")
        eng = engine_for(cfg, sd)
        set_batch(eng, batch)
        out = []
        for order in (list(reversed(range(T - n, T))), list(reversed(range(n)))):      # s = 499 .. 460, then s = 39 .. 0
            x, h, tx, th = eng.sample(eng.coef_array(coef, order), n, noise, trajectory=True)
            torch.cuda.synchronize()
            eng.sample_status()
            out += [x.cpu(), h.cpu(), tx.cpu(), th.cpu()]
            skipped[(form, order[0])] = eng.kernel_family(cfg.n_convs + 2)
        res[form] = out
    for a, b in zip(res["spec"], res["plain"]):
        assert torch.equal(a, b)
    assert skipped[("spec", n - 1)] == 1 and skipped[("plain", n - 1)] == 0


@pytest.mark.parametrize("seed", [11, 12])
def test_work_done_ahead_random_shapes_bitwise(seed, monkeypatch):
    """The rows computed ahead (whole regions and kept prefixes of changed ones: BuildParams::pa_same) and the edge records over random
    batch shapes -- 1..9 graphs, 1..12 centers, 1..9 neighbours, pockets of 24..300 atoms -- 16 steps from the noisy end of a bounded
    schedule and 16 from its quiet end: equal to PFDYN_NO_PA_SPEC=1 + PFDYN_EDGE_REC=0 bit for bit, trajectories included."""
    import random
    rng = random.Random(seed)
    for trial in range(3):
        B = rng.choice([1, 2, 5, 9])
        n_prot = [rng.choice([24, 60, 96, 200, 256, 300]) for _ in range(B)]
        n_pharm = [rng.choice([1, 2, 4, 6, 12]) for _ in range(B)]
        cfg = O.DynamicsConfig(pf_k=rng.choice([1, 3, 5, 9]))
        sd = O.make_state_dict(cfg, 60 + seed)
        batch = O.synthetic_batch([700 + 20 * seed + 7 * trial + i for i in range(B)], n_prot, n_pharm, cfg)
        Nf = int(batch.pharm_ptr[-1])
        T, n = 500, 16
        noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(seed * 10 + trial))
        coef = O.step_coefficients(O.gamma_table(T, 0.25), T)
        res = {}
        for form in ("ahead", "plain"):
            if form == "plain":
                monkeypatch.setenv("PFDYN_NO_PA_SPEC", "1")
                monkeypatch.setenv("PFDYN_EDGE_REC", "0")
            eng = engine_for(cfg, sd)
            if form == "plain":
                monkeypatch.delenv("PFDYN_NO_PA_SPEC")
                monkeypatch.delenv("PFDYN_EDGE_REC")
            set_batch(eng, batch)
            out = []
            for order in (list(reversed(range(T - n, T))), list(reversed(range(n)))):
                x, h, tx, th = eng.sample(eng.coef_array(coef, order), n, noise, trajectory=True)
                torch.cuda.synchronize()
                eng.sample_status()
                out += [x.cpu(), h.cpu(), tx.cpu(), th.cpu()]
            res[form] = out
        for a, b in zip(res["ahead"], res["plain"]):
            assert torch.equal(a, b), f"seed {seed} trial {trial}: atoms {n_prot}, centers {n_pharm}, pf_k {cfg.pf_k}"


def test_edge_records_equal_the_chased_descriptors(monkeypatch):
    """Edge records (BuildParams::rec / FusedParams::rec): the merged launch's update + build leaves, per ff / pf slot, what the next
    call's fused launch otherwise collects in two dependent round trips (end points -> the source's in-edge descriptors, element
    type, both coordinates).  Same values from another place: a run equals PFDYN_EDGE_REC=0 bit for bit -- uniform batch (the
    arithmetic-tiling kernel that reads the records), 40 noisy and 40 quiet steps of the bounded schedule, trajectories included --
    and matches the oracle's trajectory end."""
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 6)
    batch = O.synthetic_batch([840 + i for i in range(6)], [96] * 6, [6] * 6, cfg)
    Nf = int(batch.pharm_ptr[-1])
    T, n = 500, 40
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(31))
    coef = O.step_coefficients(O.gamma_table(T, 0.25), T)
    res = {}
    for form in ("rec", "plain"):
        if form == "plain":
            monkeypatch.setenv("PFDYN_EDGE_REC", "0")
        eng = engine_for(cfg, sd)
        set_batch(eng, batch)
        out = []
        for order in (list(reversed(range(T - n, T))), list(reversed(range(n)))):
            x, h, tx, th = eng.sample(eng.coef_array(coef, order), n, noise, trajectory=True)
            torch.cuda.synchronize()
            eng.sample_status()
            out += [x.cpu(), h.cpu(), tx.cpu(), th.cpu()]
        res[form] = out
    for a, b in zip(res["rec"], res["plain"]):
        assert torch.equal(a, b)


def test_fused_launch_arithmetic_tiling_equals_the_work_list_form(monkeypatch):
    """The conv-layer-0 edge launch and the fused launch of a batch whose graphs all have the same number of centers map items to
    (etype, graph, group) by arithmetic on preloaded scalars (k_n16_edge_u / k_n16_fused_u: regions at a fixed stride, groups beyond
    a region's count leave at once, the edge slots requested with the wave's start); every other batch -- and PFDYN_FUSED_UNI=0 --
    walks the work lists (k_n16_edge<true> / k_n16_fused).  Same items, same code behind the map: three denoising
    steps of a uniform batch agree bit for bit, edge lists included, and with the oracle within the single-call tolerance."""
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 5)
    batch = O.synthetic_batch([810 + i for i in range(5)], 96, [6] * 5, cfg)
    Nf = int(batch.pharm_ptr[-1])
    T, n = 100, 3
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(23))
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    out = {}
    for form in ("1", "0"):
        monkeypatch.setenv("PFDYN_FUSED_UNI", form)
        eng = engine_for(cfg, sd)
        set_batch(eng, batch)
        arr = eng.coef_array(coef, [40, 39, 38])
        eng.sample_begin(noise[0])
        st = []
        for i in range(n):
            eng.denoise_step(arr[i], noise[i + 1])
            assert eng.kernel_family(1) == 17                       # the fused launch ran
            x, h = eng.sample_frame()
            st.append((x.cpu(), h.cpu(), [tuple(t.clone() for t in eng.get_edges(et)) for et in range(3)]))
        out[form] = st
    for i in range(n):
        assert torch.equal(out["1"][i][0], out["0"][i][0]) and torch.equal(out["1"][i][1], out["0"][i][1]), i
        for et in range(3):
            assert torch.equal(out["1"][i][2][et][0], out["0"][i][2][et][0]) and torch.equal(out["1"][i][2][et][1], out["0"][i][2][et][1])
    # and against the oracle (the reference's semantics), first step
    bidx = batch.batch_idxs()
    init_com = O.segment_mean(batch.prot_x, batch.prot_ptr)
    px = batch.prot_x - init_com[bidx["prot"]]
    x_t, h_t = noise[0][:, :3].clone(), noise[0][:, 3:].clone()
    with torch.no_grad():
        px, x_t, h_t = O.sample_step(sd, cfg, batch, coef, 40, px, x_t, h_t, noise[1][:, :3], noise[1][:, 3:])
    ox = x_t - O.segment_mean(px, batch.prot_ptr)[bidx["pharm"]] + init_com[bidx["pharm"]]
    torch.testing.assert_close(out["1"][0][0], ox, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out["1"][0][1], h_t, rtol=1e-3, atol=1e-3)


def test_exchange_timeout_surfaces_on_the_same_run_and_the_handle_recovers():
    """The merged last launch (k_rg_node_hs_build) hands eps over through polled exchange words with a bounded poll; a hand-over
    that never arrives must make THAT run fail, not the next one.  pf_debug_xchg_fault makes the
    producers skip one word (and shortens the poll): pf_sample returns, the caller waits for x_0 as it always does, and
    pf_sample_status reports PF_ERR_EXCHANGE for this run; the handle switches to the separate launches, its next run starts from
    re-armed words and equals the PFDYN_HS_BUILD=0 result bit for bit; the status is then clean again."""
    from pharmacoforge_amd import PfError
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 3)
    batch = O.synthetic_batch([700 + i for i in range(4)], 64, [4, 6, 5, 6], cfg)
    Nf = int(batch.pharm_ptr[-1])
    T, n = 100, 4
    noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(17))
    coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    arr = eng.coef_array(coef, [40, 39, 38, 37])
    x_ok, h_ok = eng.sample(arr, n, noise)
    assert eng.kernel_family(cfg.n_convs) == 2                      # the merged launch is what ran
    torch.cuda.synchronize()
    eng.sample_status()                                             # clean
    eng.xchg_fault(True, poll_max=64)
    x_bad, h_bad = eng.sample(arr, n, noise)
    torch.cuda.synchronize()                                        # (what a caller does before reading x_0)
    with pytest.raises(PfError, match="time-out"):
        eng.sample_status()
    assert eng.xchg_timeouts() >= 1
    eng.sample_status()                                             # reported once
    eng.xchg_fault(False)
    x2, h2 = eng.sample(arr, n, noise)                              # the same handle: separate launches now, words re-armed by begin
    assert eng.kernel_family(cfg.n_convs) == 0
    torch.cuda.synchronize()
    eng.sample_status()
    import os
    os.environ["PFDYN_HS_BUILD"] = "0"
    try:
        ref = engine_for(cfg, sd)
    finally:
        del os.environ["PFDYN_HS_BUILD"]
    set_batch(ref, batch)
    xr, hr = ref.sample(arr, n, noise)
    assert torch.equal(x2, xr) and torch.equal(h2, hr)
    torch.testing.assert_close(x_ok, xr, rtol=2e-3, atol=2e-3)      # (merged launch + center hoist vs separate launches: summation order)
    # a caller that never asks is told by the next begin on the handle
    eng2 = engine_for(cfg, sd)
    set_batch(eng2, batch)
    eng2.xchg_fault(True, poll_max=64)
    eng2.sample(arr, n, noise)
    torch.cuda.synchronize()
    eng2.xchg_fault(False)
    with pytest.raises(PfError, match="time-out"):
        eng2.sample(arr, n, noise)
    x3, h3 = eng2.sample(arr, n, noise)
    assert torch.equal(x3, xr) and torch.equal(h3, hr)


def test_exchange_words_canonicalise_an_all_ones_nan():
    """An all-ones NaN is the exchange's 'not yet written' pattern; hardware propagates NaN payloads from inputs, so the producers
    canonicalise it (pf_xchg_word): a batch whose noise carries that NaN ends in NaN results, not in a time-out."""
    cfg = O.DynamicsConfig()
    sd = O.make_state_dict(cfg, 3)
    batch = O.synthetic_batch([700], 64, [4], cfg)
    noise = torch.randn(3, 4, 9, generator=torch.Generator().manual_seed(1))
    noise[0, 0, 3] = torch.tensor([-1], dtype=torch.int32).view(torch.float32)[0]
    assert noise[0].view(torch.int32)[0, 3] == -1
    coef = O.step_coefficients(O.gamma_table(100, 1e-5), 100)
    eng = engine_for(cfg, sd)
    set_batch(eng, batch)
    x, h = eng.sample(eng.coef_array(coef, [40, 39]), 2, noise)
    torch.cuda.synchronize()
    eng.sample_status()
    assert eng.xchg_timeouts() == 0 and torch.isnan(h).any()


def test_split_bf16x6_variant_library_passes_the_chain_and_trajectory_goldens():
    """libpfdyn_split.so (make split; pf_device.h N16_SPLIT): the n16 blocks' 128-input scalar Linears on bf16 matrix instructions with
    both operands split into three bf16 planes and fp32 accumulation -- an A/B variant, never the default (profiles/r05/
    split_bf16x6_ab.txt).  Its claim is fp32-class arithmetic, so it must pass the same tests at the SAME tolerances: here the n16
    chain tests (reference modules, oracle) and the reference's trajectories, in a child process that loads the variant through
    PFDYN_LIB.  (The whole n16 / parity / full-size suites passed under it when it was built.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "pharmacophore-diffusion_amd", "csrc", "libpfdyn_split.so")
    if not os.path.exists(lib):
        pytest.skip("libpfdyn_split.so has not been built (make -C pharmacophore-diffusion_amd/csrc split)")
    env = dict(os.environ, PFDYN_LIB=lib)
    sel = ("test_n16_chain_units_vs_reference_modules or test_n16_every_chain_vs_oracle or test_trajectory_vs_golden or test_bounded_T500 or "
           "test_fused_launch_arithmetic_tiling or test_dynamics_vs_golden_and_oracle")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(root, "tests", "test_gpu_n16.py"),
                        os.path.join(root, "tests", "test_gpu_parity.py"), "-k", sel], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert " passed" in r.stdout and "no tests ran" not in r.stdout
