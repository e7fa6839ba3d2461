import sys, os, ctypes
os.environ["PFDYN_LIB"] = os.path.join(os.path.dirname(__file__), "..", "pharmacophore-diffusion_amd", "csrc", "variants", "libpfdyn_tstamps.so")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import pf_oracle as O
import test_gpu_train as T
B = 256
cfg = O.DynamicsConfig()
sizes = [4 + (i % 5) for i in range(B)]
batch = O.synthetic_batch(list(range(B)), 256, sizes, cfg)
sd = O.make_state_dict(cfg, 0)
eng = T.make_engine(cfg, sd, batch)
Nf = int(batch.pharm_ptr[-1])
g = torch.Generator().manual_seed(0)
x = torch.randn(Nf, 3, generator=g).cuda(); h = torch.randn(Nf, 6, generator=g).cuda(); t = torch.rand(B, generator=g).cuda()
gh = torch.randn(Nf, 6, generator=g).cuda() * 1e-3; gx = torch.randn(Nf, 3, generator=g).cuda() * 1e-3
lib = eng.lib
buf = (ctypes.c_ulonglong * 128)()
for it in range(2):
    eng.train_forward(x, h, t, dropout=0.1, seed=1)
    torch.cuda.synchronize(); lib.pft_read_stamps(buf, 1)
    eng.train_backward(gh, gx); torch.cuda.synchronize()
    n = lib.pft_read_stamps(buf, 1)
print("stamps", n)
prev = None
for i in range(n):
    v = buf[i]; idn = v >> 48; cyc = v & 0xffffffffffff
    print(f"{idn:3d} +{(cyc - prev) if prev is not None else 0}")
    prev = cyc
