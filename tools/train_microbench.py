#!/usr/bin/env python3
"""Forward / backward device time of one training step through the C ABI (no Python model, no optimiser):
    python tools/train_microbench.py [batch]          # B x 256-atom pockets, 4..8 centers, dropout 0.1
With a -DPFT_STAMPS build (csrc/build_variant.sh NAME "-DPFT_STAMPS -DPFT_STAMP_MIN=30 -DPFT_STAMP_BLOCK=100") and
PFDYN_LIB pointing at it, `--stamps` prints the cycle stamps the gradient kernels recorded at their phase boundaries."""
import ctypes
import itertools
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pharmacoforge_amd as pfa
from pharmacoforge_amd import synthetic

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 256
dev = torch.device("cuda", 0)
eng = pfa.PfEngine(device=dev)
eng.load_state_dict(synthetic.make_state_dict(0))
xs, hs = zip(*[synthetic.synthetic_pocket(i, 256) for i in range(B)])
prot_x, prot_h = torch.cat(xs).to(dev), torch.cat(hs).to(dev)
prot_ptr = torch.arange(B + 1, dtype=torch.int64) * 256
sizes = [4 + (i % 5) for i in range(B)]
pharm_ptr = torch.tensor([0] + list(itertools.accumulate(sizes)), dtype=torch.int64)
pp_src, pp_dst = eng.build_pp_edges(prot_x, prot_ptr)
eng.set_batch(prot_x, prot_h, prot_ptr, pharm_ptr, pp_src, pp_dst)
Nf = int(pharm_ptr[-1])
g = torch.Generator().manual_seed(0)
x = torch.randn(Nf, 3, generator=g).to(dev); h = torch.randn(Nf, 6, generator=g).to(dev); t = torch.rand(B, generator=g).to(dev)
gh = torch.randn(Nf, 6, generator=g).to(dev) * 1e-3; gx = torch.randn(Nf, 3, generator=g).to(dev) * 1e-3
for _ in range(2):
    eng.train_forward(x, h, t, dropout=0.1, seed=1); eng.train_backward(gh, gx)
torch.cuda.synchronize()
if "--stamps" in sys.argv:
    buf = (ctypes.c_ulonglong * 128)()
    eng.lib.pft_read_stamps(buf, 1)
    eng.train_forward(x, h, t, dropout=0.1, seed=1); eng.train_backward(gh, gx); torch.cuda.synchronize()
    n, prev = eng.lib.pft_read_stamps(buf, 1), None
    for i in range(n):
        idn, cyc = buf[i] >> 48, buf[i] & 0xffffffffffff
        print(f"{idn:3d} +{(cyc - prev) if prev is not None else 0}")
        prev = cyc
    sys.exit(0)
K = 5
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(K):
    e[0].record(); eng.train_forward(x, h, t, dropout=0.1, seed=1); e[1].record(); eng.train_backward(gh, gx); e[2].record()
    torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
print(f"B={B} forward {tf / K:.3f} ms  backward {tb / K:.3f} ms  -> {B / ((tf + tb) / K) * 1e3:.0f} graphs/s")
