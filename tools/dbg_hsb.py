import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import pf_oracle as O
import pharmacoforge_amd as pfa
cfg = O.DynamicsConfig(n_convs=3, n_message_gvps=2, n_update_gvps=3, n_noise_gvps=2)
sd = O.make_state_dict(cfg, 3)
n_prot, n_pharm = [3, 70, 12, 48, 7, 33], [1, 10, 3, 8, 2, 5]
batch = O.synthetic_batch([700 + i for i in range(len(n_prot))], n_prot, n_pharm, cfg)
Nf = int(batch.pharm_ptr[-1])
T, n = 100, 3
noise = torch.randn(n + 1, Nf, 9, generator=torch.Generator().manual_seed(17))
coef = O.step_coefficients(O.gamma_table(T, 1e-5), T)
out = {}
for hsb in ("0", "1", "1b"):
    os.environ["PFDYN_HS_BUILD"] = hsb[0]
    os.environ["PFDYN_N16"] = "7"
    eng = pfa.PfEngine(pharm_nf=cfg.pharm_nf, rec_nf=cfg.rec_nf, n_convs=cfg.n_convs, n_message_gvps=cfg.n_message_gvps, n_update_gvps=cfg.n_update_gvps,
                       n_noise_gvps=cfg.n_noise_gvps, message_norm=cfg.message_norm, ff_k=cfg.ff_k, pf_k=cfg.pf_k,
                       graph_cutoffs={"pp": cfg.cutoff_pp, "pf": cfg.cutoff_pf, "fp": cfg.cutoff_fp, "ff": cfg.cutoff_ff})
    eng.load_state_dict(sd)
    eng.set_batch(batch.prot_x, batch.prot_h, batch.prot_ptr, batch.pharm_ptr, batch.pp_src, batch.pp_dst)
    arr = eng.coef_array(coef, [40, 39, 38])
    eng.sample_begin(noise[0])
    st = []
    for i in range(n):
        eng.denoise_step(arr[i], noise[i + 1])
        fam = eng.kernel_family(cfg.n_convs)
        x, h = eng.sample_frame()
        eh, ex = eng.last_eps()
        st.append((x.cpu(), h.cpu(), eh.cpu(), ex.cpu()))
    out[hsb] = st
    print(hsb, "family", fam, "timeouts", eng.xchg_timeouts())
for i in range(n):
    for q in (0, 1, 2, 3):
        d = (out["1"][i][q] - out["0"][i][q]).abs()
        d2 = (out["1"][i][q] - out["1b"][i][q]).abs()
        print("step", i, ("x", "h", "eps_h", "eps_x")[q], "max diff merged-vs-separate", float(d.max()), "rows", d.max(1).values.tolist() if i == 0 else "", "merged repeat", float(d2.max()))
d = out["1"][0][0] - out["0"][0][0]
torch.set_printoptions(precision=7, linewidth=200)
print("delta x rows 0..11:\n", d[:12])
print("eps_x rows 0..11:\n", out["1"][0][3][:12])
print("sum of delta over each 4-row item:", [d[4*k:4*k+4].sum(0).tolist() for k in range(3)])
print("sum over graph 1 (rows 1..10):", d[1:11].sum(0).tolist())
