"""bf16 training leg vs the fp32 path on the GPU: per-tensor cosine / relative error of every parameter gradient, outputs, and
the time of forward + backward in both modes.  usage: python3 tools/bf16_grad_check.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pharmacoforge_amd as pfa
from pharmacoforge_amd import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
eng = pfa.PfEngine(device=dev)
eng.load_state_dict(synthetic.make_state_dict(0))
sizes = [4 + (i % 5) for i in range(B)]
pockets = [synthetic.synthetic_pocket(900 + i, 256) for i in range(B)]
prot_x = torch.cat([p[0] for p in pockets]); prot_h = torch.cat([p[1] for p in pockets])
prot_ptr = torch.arange(B + 1, dtype=torch.int64) * 256
pharm_ptr = torch.tensor([0] + list(__import__("itertools").accumulate(sizes)), dtype=torch.int64)
pp_src, pp_dst = eng.build_pp_edges(prot_x.to(dev), prot_ptr)
eng.set_batch(prot_x, prot_h, prot_ptr, pharm_ptr, pp_src, pp_dst)
Nf = int(pharm_ptr[-1])
gen = torch.Generator().manual_seed(5)
com = torch.stack([p[0].mean(0) for p in pockets])
gid_p = torch.repeat_interleave(torch.arange(B), 256)
gid_f = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
px = prot_x - com[gid_p]
x_t = 2.5 * torch.randn(Nf, 3, generator=gen)
h_t = torch.randn(Nf, 6, generator=gen)
t = torch.rand(B, generator=gen)
w_h, w_x = torch.randn(Nf, 6, generator=gen) / Nf, torch.randn(Nf, 3, generator=gen) / Nf
res = {}
for mode in ("f32", "bf16"):
    eng.set_train_precision(mode)
    assert eng.train_precision() == mode
    eh, ex = eng.train_forward(x_t, h_t, t, prot_x=px, dropout=0.1, seed=4242)
    g = eng.train_backward(w_h, w_x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.train_forward(x_t, h_t, t, prot_x=px, dropout=0.1, seed=4242)
        eng.train_backward(w_h, w_x)
    torch.cuda.synchronize()
    res[mode] = (eh.cpu(), ex.cpu(), g.cpu(), (time.perf_counter() - t0) / 10)
    print(f"{mode}: forward + backward {res[mode][3] * 1e3:.3f} ms, finite {bool(torch.isfinite(g).all())}")
eh0, ex0, g0, _ = res["f32"]; eh1, ex1, g1, _ = res["bf16"]
print("eps_h rel", float((eh1 - eh0).norm() / eh0.norm()), "eps_x rel", float((ex1 - ex0).norm() / ex0.norm()))
rows = []
for name, off, n in eng.param_layout():
    if n == 0: continue
    a, b = g0[off:off + n].double(), g1[off:off + n].double()
    na = float(a.norm())
    if na == 0: continue
    cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
    rel = float((a - b).norm() / na)
    rows.append((cos, rel, name, n, na))
rows.sort()
print("worst cosine:")
for r in rows[:12]: print(f"  cos {r[0]:.6f} rel {r[1]:.4f} n {r[3]:6d} |g| {r[4]:.3e} {r[2]}")
print("worst rel:")
for r in sorted(rows, key=lambda r: -r[1])[:12]: print(f"  cos {r[0]:.6f} rel {r[1]:.4f} n {r[3]:6d} |g| {r[4]:.3e} {r[2]}")
print("tensors", len(rows), "min cos", rows[0][0], "max rel", max(r[1] for r in rows), "whole-vector cos",
      float((g0.double() * g1.double()).sum() / (g0.double().norm() * g1.double().norm())), "whole rel", float((g0 - g1).norm() / g0.norm()))
