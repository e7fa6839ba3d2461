"""CPU tests of the host side: the C-ABI library loads and exports every symbol of include/pfdyn.h
(no compute calls), schedule algebra, synthetic generators, batch bookkeeping, state-dict /
checkpoint layout, xyz writer, metrics and the 2-rank (gloo) reduction path."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import pharmacoforge_amd as pfa
from oracle import pf_oracle as O
from helpers import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_DYN = dict(vector_size=16, n_convs=2, n_hidden_scalars=128, message_norm='mean', dropout=0.1, ff_k=0, pf_k=5,
               n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4)
DEV_GRAPH = {'graph_cutoffs': {'pp': 3.5, 'pf': 8, 'fp': 8, 'ff': 9}}


def make_model(T=100):
    return pfa.PharmacophoreDiff(6, 11, pfa.analysis.ph_idx_to_type, None, n_timesteps=T, graph_config=DEV_GRAPH,
                                 dynamics_config=DEV_DYN, precision=1e-5)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pfdyn.h")).read()
    declared = set(re.findall(r"\b(pf_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"pf_status"}
    assert len(declared) >= 18
    lib = ctypes.CDLL(pfa._lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libpfdyn.so does not export {name}"
    assert declared == set(pfa._lib.SYMBOLS), declared ^ set(pfa._lib.SYMBOLS)
    assert pfa._lib.load().pf_version().startswith(b"libpfdyn")


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    lib = pfa._lib.load()
    cfg = pfa._lib.PfConfig(pfa._lib.PF_ABI_VERSION, 6, 11, 16, 128, 2, 3, 2, 4, 0, 1.0, 0, 5, 3.5, 8, 8, 9, 15.0, 16)
    h = ctypes.c_void_p()
    rc = lib.pf_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -2 and b"no HIP device" in lib.pf_last_error(None)        # PF_ERR_HIP, loudly
    with pytest.raises(pfa.PfError):
        pfa.PfEngine()
    m = make_model()
    with pytest.raises(RuntimeError):
        m.dynamics.engine()


def test_bad_config_rejected():
    lib = pfa._lib.load()
    cfg = pfa._lib.PfConfig(pfa._lib.PF_ABI_VERSION, 6, 11, 8, 128, 2, 3, 2, 4, 0, 1.0, 0, 5, 3.5, 8, 8, 9, 15.0, 16)
    h = ctypes.c_void_p()
    assert lib.pf_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and b"vector_size" in lib.pf_last_error(None)


@pytest.mark.parametrize("T", [50, 500])
def test_schedule_module_matches_reference_tables(T):
    z = load("schedule.npz")
    tag = f"T{T}_p1e-05"
    sched = pfa.PredefinedNoiseSchedule('polynomial_2', T, 1e-5)
    assert torch.equal(sched.gamma.detach(), z["gamma_" + tag])
    c = pfa.schedule.step_coefficients(sched.gamma, T)
    # the gamma table is a float64 numpy pipeline (bit-exact everywhere); the per-step coefficients are fp32 torch
    # expm1 / softplus / exp, whose vectorised CPU kernels differ by an ulp between instruction sets (the goldens were
    # recorded on this container's CPU; the GPU box's host gives 1-ulp differences)
    for k, zk in (("alpha_t_given_s", "a_ts_"), ("var_terms", "var_"), ("sigma", "sigma_")):
        torch.testing.assert_close(c[k], z[zk + tag], rtol=1e-6, atol=1e-9)
    t = torch.tensor([0.0, 0.5, 1.0])
    assert torch.equal(sched(t), sched.gamma[torch.tensor([0, T // 2, T])])


def test_synthetic_generators_agree_with_oracle_copies():
    x1, h1 = pfa.synthetic.synthetic_pocket(7, 64)
    x2, h2 = O.synthetic_pocket(7, 64)
    assert torch.equal(x1, x2) and torch.equal(h1, h2)
    a, b = pfa.synthetic.make_state_dict(3), O.make_state_dict(O.DynamicsConfig(), 3)
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def test_state_dict_layout_and_checkpoint_roundtrip(tmp_path):
    m = make_model()
    sd = m.state_dict()
    assert len(sd) == 245 and "gamma.gamma" in sd
    assert sd["dynamics.noise_predictor.conv_layers.0.dropout.vector_dropout.dummy_param"].shape == (0,)
    ref = O.make_state_dict(O.DynamicsConfig(), 0)
    full = dict(ref); full["gamma.gamma"] = sd["gamma.gamma"]
    m.load_state_dict(full, strict=True)                               # reference key set loads 1:1
    p = tmp_path / "last.ckpt"
    m.save_checkpoint(p)
    m2 = pfa.PharmacophoreDiff.load_from_checkpoint(p)
    assert all(torch.equal(v, m2.state_dict()[k]) for k, v in m.state_dict().items())
    assert m2.n_timesteps == 100 and m2.dynamics.pf_k == 5


def test_model_from_config_dev_yml():
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "golden", "dev_config_subset.yml")))
    m = pfa.model_from_config(cfg)
    assert m.n_timesteps == 100 and m.n_pharm_feats == 6 and m.n_prot_feats == 11
    assert len(m.dynamics.noise_predictor.conv_layers) == 2


def pocket(seed, n_prot, n_pharm):
    cfg = O.DynamicsConfig()
    b = O.synthetic_batch([seed], n_prot, n_pharm, cfg)
    return pfa.PocketGraph(b.prot_x, b.prot_h, b.prot_ptr, b.pharm_ptr, b.pp_src, b.pp_dst,
                           torch.zeros(n_pharm, 3), torch.zeros(n_pharm, 6))


def test_batch_unbatch_copy_graph():
    g1, g2 = pocket(1, 20, 3), pocket(2, 30, 5)
    g = pfa.batch([g1, g2])
    assert g.batch_size == 2 and g.num_nodes("prot") == 50 and g.batch_num_nodes("pharm").tolist() == [3, 5]
    bi = g.batch_idxs()
    assert bi["prot"].tolist() == [0] * 20 + [1] * 30 and bi["pharm"].tolist() == [0] * 3 + [1] * 5
    u = pfa.unbatch(g)
    assert torch.equal(u[1].prot_x, g2.prot_x) and torch.equal(u[1].pp_src, g2.pp_src) and torch.equal(u[1].pp_dst, g2.pp_dst)
    copies = pfa.copy_graph(g1, 3, pharm_feats_per_copy=torch.tensor([4, 8, 3]))
    assert [c.num_nodes("pharm") for c in copies] == [4, 8, 3] and all(c.pharm_h0.abs().sum() == 0 for c in copies)
    assert copies[0].prot_x.data_ptr() != g1.prot_x.data_ptr() and torch.equal(copies[2].prot_x, g1.prot_x)


def test_index_arrays_i32_are_cached_per_graph_and_follow_the_tensors():
    """PocketGraph.index_arrays_i32: made at collate time (batch()), shared with .to() copies, rebuilt when an index tensor is
    replaced or written in place."""
    cfg = O.DynamicsConfig()
    parts = [pfa.graph.PocketGraph(b.prot_x, b.prot_h, b.prot_ptr, b.pharm_ptr, b.pp_src, b.pp_dst)
             for b in (O.synthetic_batch([3], 20, 2, cfg), O.synthetic_batch([4], 16, 3, cfg))]
    g = pfa.graph.batch(parts)
    assert "_i32_cache" in g.__dict__                      # made by batch()
    a = g.index_arrays_i32()
    assert all(x.dtype == np.int32 and x.flags["C_CONTIGUOUS"] for x in a)
    assert np.array_equal(a[0], g.prot_ptr.numpy()) and np.array_equal(a[2], g.pp_src.numpy()) and np.array_equal(a[3], g.pp_dst.numpy())
    assert g.index_arrays_i32()[2] is a[2]                 # cached
    assert g.to("cpu").index_arrays_i32()[2] is a[2]       # carried by to()
    g.pp_src[0] = g.pp_src[0]                              # an in-place write bumps the version: rebuilt
    assert g.index_arrays_i32()[2] is not a[2]
    g.pp_src = g.pp_src.clone()
    b2 = g.index_arrays_i32()
    assert np.array_equal(b2[2], a[2])


def test_xyz_writer_matches_reference_output():
    z = load("traj_c1.npz")
    g = pocket(0, 64, 4)
    g.pharm_x0, g.pharm_h0 = z["x0"], z["h0"]
    ph = pfa.SampledPharmacophore(g, pfa.analysis.ph_idx_to_type, traj_frames=(z["pos_frames"], z["feat_frames"]))
    assert ph.to_xyz_file() == str(z["xyz"])
    assert ph.traj_to_xyz().count("\n") == 51 * 5
    txt = pfa.write_pharmacophore_file([z["x0"]], [z["h0"].argmax(dim=1).tolist()], pfa.analysis.ph_idx_to_type)
    assert txt == str(z["xyz"])


def test_validity_metric():
    g = pocket(3, 10, 2)
    g.pharm_x0 = torch.tensor([[0., 0, 0], [10., 0, 0]])
    g.pharm_h0 = torch.eye(6)[[1, 5]]                       # HydrogenDonor, Hydrophobic
    g.prot_ph_x = torch.tensor([[3.9, 0, 0], [10., 5.1, 0]])
    g.prot_ph_h = torch.eye(6)[[2, 5]]                      # acceptor within 4 A; hydrophobic just outside 5 A
    ph = pfa.SampledPharmacophore(g, pfa.analysis.ph_idx_to_type)
    assert pfa.SampleAnalyzer().analyze([ph]) == {'validity': 0.5}
    assert pfa.SampleAnalyzer().pharm_feat_freq([ph]).tolist() == [0, 1, 0, 0, 0, 1]


def test_metrics_match_the_reference_analyzer():
    """analysis.SampleAnalyzer / compute_complementarity against tests/golden/metrics.npz -- the reference's own
    SampleAnalyzer.analyze, pharm_feat_freq and compute_complementarity(return_count=True) (analysis/metrics.py:9-86) run on its own
    SampledPharmacophore objects: 24 seeded samples plus the edges of the rule (a center exactly at the matching distance, just
    beyond it, complementary-but-far next to close-but-not-complementary, argmax ties).  Counts per sample exact, validity to the
    reference's float32 division."""
    z = load("metrics.npz")
    fp, rp = z["pharm_ptr"].tolist(), z["prot_ph_ptr"].tolist()
    samples, counts = [], []
    for i in range(int(z["n"])):
        g = pocket(3, 5, fp[i + 1] - fp[i])
        g.pharm_x0, g.pharm_h0 = z["pharm_x"][fp[i]:fp[i + 1]], z["pharm_h"][fp[i]:fp[i + 1]]
        g.prot_ph_x, g.prot_ph_h = z["prot_ph_x"][rp[i]:rp[i + 1]], z["prot_ph_h"][rp[i]:rp[i + 1]]
        ph = pfa.SampledPharmacophore(g, pfa.analysis.ph_idx_to_type)
        samples.append(ph)
        rt = [pfa.analysis.ph_idx_to_type[int(k)] for k in g.prot_ph_h.argmax(dim=1)]
        counts.append(int(pfa.analysis.compute_complementarity(ph.ph_types, ph.ph_coords, rt, g.prot_ph_x, return_count=True)))
    assert counts == z["counts"].tolist()
    an = pfa.SampleAnalyzer()
    assert abs(an.analyze(samples)["validity"] - float(z["validity"])) < 1e-7
    assert abs(an.analyze(samples[:len(samples) // 2])["validity"] - float(z["validity_first_half"])) < 1e-7
    assert an.pharm_feat_freq(samples).tolist() == z["freq"].tolist()


def _rank_fn(rank, world, port, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = pocket(3, 10, 2)
    g.pharm_x0 = torch.tensor([[0., 0, 0], [10., 0, 0]])
    g.pharm_h0 = torch.eye(6)[[1, 5]]
    g.prot_ph_x = torch.tensor([[3.9, 0, 0], [10., 4.9 if rank else 5.1, 0]])
    g.prot_ph_h = torch.eye(6)[[2, 5]]
    ph = pfa.SampledPharmacophore(g, pfa.analysis.ph_idx_to_type)
    res = pfa.SampleAnalyzer().analyze([ph], process_group=dist.group.WORLD)
    freq = pfa.SampleAnalyzer().pharm_feat_freq([ph], process_group=dist.group.WORLD)
    q.put((rank, res['validity'], freq.tolist()))
    dist.destroy_process_group()


def test_metrics_allreduce_two_ranks_gloo():
    """The N>1 path: graphs are dealt round-robin over ranks (no data-path collective) and only the
    validity numerator/denominator + type counts are all-reduced."""
    import torch.multiprocessing as mp
    idx = [list(range(r, 7, 2)) for r in range(2)]
    assert sorted(idx[0] + idx[1]) == list(range(7))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_fn, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(30) for p in procs]
    assert out[0][1] == out[1][1] == 0.75            # (1 + 2) valid of (2 + 2) centers over both ranks
    assert out[0][2] == [0, 2, 0, 0, 0, 2]


def _grad_rank_fn(rank, world, port, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(0)
    m = make_model()                                  # CPU parameters: only the gradient bookkeeping is exercised
    for i, p in enumerate(m.dynamics.parameters()):
        if p.numel():
            p.grad = torch.full_like(p, float(rank + 1) * (1 + i % 3))
    flat = m.dynamics.allreduce_gradients(average=True)
    got = [float(p.grad.reshape(-1)[0]) for p in m.dynamics.parameters() if p.numel()][:6]
    n = sum(p.numel() for p in m.dynamics.parameters())
    q.put((rank, got, flat.numel() == n))
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo():
    """Data-parallel training path: one all-reduce of the flat gradient vector; every parameter's .grad becomes the
    rank average."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    procs = [ctx.Process(target=_grad_rank_fn, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(30) for p in procs]
    expect = [1.5 * (1 + i % 3) for i in range(6)]
    assert out[0][1] == out[1][1] == expect
    assert out[0][2] and out[1][2]


def test_product_code_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it."""
    offenders = []
    targets = [os.path.join(ROOT, f) for f in ("generate_pharmacophores.py", "train.py", "test.py", "pharmacoforge_amd.py")]
    for base in ("pharmacophore-diffusion_amd", "tools", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            targets += [os.path.join(dp, f) for f in files if f.endswith((".py", ".cpp", ".hip", ".h"))]
    for path in targets:
        text = open(path, errors="ignore").read()
        if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or re.search(r"^[^#/\n]*\bpf_oracle\b", text, re.M):
            offenders.append(os.path.relpath(path, ROOT))
    assert not offenders, offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"\boracle\b", bench)]
    start = bench.index("def cpu_baseline")
    assert all(u > start or "import" not in bench[max(0, u - 40):u] for u in uses)


def _skewed_edge_counts(n, seed=0):
    """pp edge counts of pockets whose atom counts spread 2-3x (150..450 atoms at ~6.8 edges per atom)."""
    rng = np.random.default_rng(seed)
    return [int(6.8 * a) for a in rng.integers(150, 451, size=n)]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_shard_by_work_balances_skewed_pockets(world):
    from pharmacoforge_amd.sharding import shard_by_work, shard_loads
    w = _skewed_edge_counts(200)
    shards = shard_by_work(w, world)
    assert sorted(i for s in shards for i in s) == list(range(200))          # a partition
    loads = shard_loads(w, shards)
    assert max(loads) / min(loads) <= 1.1, loads
    assert shards == shard_by_work(list(w), world)                            # pure function: ranks agree without talking
    # dealing by index on the same list is what this replaces
    naive = [sum(w[i] for i in range(r, 200, world)) for r in range(world)]
    assert max(loads) - min(loads) <= max(naive) - min(naive)
    assert shard_by_work([], 3) == [[], [], []] and shard_by_work([5.0], 2) == [[0], []]


def _shard_rank_fn(rank, world, port, q):
    import torch.distributed as dist
    from pharmacoforge_amd.sharding import shard_by_work
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    w = _skewed_edge_counts(64, seed=3)
    mine = shard_by_work(w, world)[rank]                       # computed locally on every rank
    load = torch.tensor([float(sum(w[i] for i in mine))], dtype=torch.float64)
    lo, hi = load.clone(), load.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    seen = torch.zeros(64)
    seen[mine] = 1
    dist.all_reduce(seen)
    q.put((rank, float(hi / lo), seen.tolist() == [1.0] * 64))
    dist.destroy_process_group()


def test_shard_by_work_two_ranks_gloo():
    """The N>1 path of the sampling drivers: every rank derives its own shard from the same weights; together the shards
    cover every pocket exactly once and the per-rank edge sums differ by at most 10 %."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + os.getpid() % 2000
    procs = [ctx.Process(target=_shard_rank_fn, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(30) for p in procs]
    assert all(ratio <= 1.1 and cover for _, ratio, cover in out), out


def test_rank_processes_take_disjoint_slices_of_the_host():
    """sharding.host_share / pin_host_threads (what every rank process of bench.py / train.py / test.py calls before touching the
    GPU): the shares of the ranks of a node are disjoint, contiguous and cover the allowed CPUs; a rank pins itself to its share
    and sizes torch's pool to it; a single-rank run is left alone.  (In a child process: the affinity change is for good.)"""
    import subprocess
    import sys
    from pharmacoforge_amd import sharding as S
    cpus = list(range(3, 259))
    shares = [S.host_share(r, 8, cpus) for r in range(8)]
    assert all(n == 32 for _, n in shares)
    assert sorted(c for sh, _ in shares for c in sh) == cpus
    assert all(sh == list(range(sh[0], sh[0] + 32)) for sh, _ in shares)
    assert S.host_share(1, 3, [0, 1])[1] == 1                       # more ranks than CPUs: one each, wrapping
    code = ("import os, json, sys; sys.path.insert(0, %r); import torch; from pharmacoforge_amd.sharding import pin_host_threads; "
            "a = pin_host_threads(0, 1); n0 = len(os.sched_getaffinity(0)); b = pin_host_threads(1, 2); "
            "print(json.dumps([a, b, n0, sorted(os.sched_getaffinity(0)), torch.get_num_threads(), os.environ['OMP_NUM_THREADS']]))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1500:]
    import json
    a, b, n0, aff, nthr, omp = json.loads(out.stdout.strip().splitlines()[-1])
    assert a["pinned"] is False and a["host_threads_per_rank"] == n0
    if n0 >= 2:
        assert b["pinned"] is True and len(aff) == n0 // 2 and b["host_threads_per_rank"] == min(n0 // 2, 16) == nthr == int(omp)
