"""Per-tensor gradient error report (debug aid)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import pf_oracle as O
from helpers import GRAD_CASES, batch_from, load
import test_gpu_train as T

name = sys.argv[1] if len(sys.argv) > 1 else "train_grads.npz"
p_drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
z = load(name); cfg = GRAD_CASES[name]; batch = batch_from(z); TT = int(z["T"])
sd = O.make_state_dict(cfg, int(z["wseed"]))
eng = T.make_engine(cfg, sd, batch)
x_t, h_t, prot_x, t = T.noised_inputs(cfg, batch, z, TT)
Np, Nf = int(batch.prot_ptr[-1]), int(batch.pharm_ptr[-1])
eps_h, eps_x = eng.train_forward(x_t, h_t, t, prot_x=prot_x, dropout=p_drop, seed=1234)
drop = T.masks_from_engine(eng, cfg, p_drop, 1234, Np, Nf) if p_drop > 0 else None
leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
with torch.enable_grad():
    oh, ox = O.dynamics_forward(leaf, cfg, batch, prot_x, x_t, h_t, t, dropout=drop)
    loss = (z["eps_x"] - ox).square().sum() / z["eps_x"].numel() + (z["eps_h"] - oh).square().sum() / z["eps_h"].numel()
    loss.backward()
print("fwd err", float((eps_h.cpu() - oh.detach()).abs().max()), float((eps_x.cpu() - ox.detach()).abs().max()))
g_h = (-2.0 / z["eps_h"].numel()) * (z["eps_h"] - eps_h.cpu())
g_x = (-2.0 / z["eps_x"].numel()) * (z["eps_x"] - eps_x.cpu())
grad = eng.train_backward(g_h, g_x)
torch.cuda.synchronize()
got = T.flat_to_dict(eng, grad)
nbad = 0
for k, v in leaf.items():
    if v.numel() == 0: continue
    r = torch.zeros_like(v) if v.grad is None else v.grad
    g = got[k].reshape(r.shape)
    sc = float(r.abs().max()); err = float((g - r).abs().max())
    flag = "" if err <= 2e-3 * sc + 1e-7 else "  <-- BAD"
    nbad += bool(flag)
    if flag or "-v" in sys.argv: print(f"{k:95s} err {err:.3e} scale {sc:.3e} got {float(g.abs().max()):.3e}{flag}")
print("bad tensors:", nbad, "of", len(leaf))
