import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import pf_oracle as O
import test_gpu_train as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = O.DynamicsConfig()
sizes = [4 + (i % 5) for i in range(B)]
batch = O.synthetic_batch(list(range(B)), 256, sizes, cfg)
sd = O.make_state_dict(cfg, 0)
eng = T.make_engine(cfg, sd, batch)
Nf = int(batch.pharm_ptr[-1])
g = torch.Generator().manual_seed(0)
x = torch.randn(Nf, 3, generator=g).cuda(); h = torch.randn(Nf, 6, generator=g).cuda(); t = torch.rand(B, generator=g).cuda()
gh = torch.randn(Nf, 6, generator=g).cuda() * 1e-3; gx = torch.randn(Nf, 3, generator=g).cuda() * 1e-3
def step():
    eng.train_forward(x, h, t, dropout=0.1, seed=1)
    return eng.train_backward(gh, gx)
for _ in range(2): step()
torch.cuda.synchronize()
K = 5
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
tf = tb = 0.0
for _ in range(K):
    e0.record(); eng.train_forward(x, h, t, dropout=0.1, seed=1); e1.record(); eng.train_backward(gh, gx); e2.record()
    torch.cuda.synchronize()
    tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
print(f"B={B} forward {tf/K:.3f} ms  backward {tb/K:.3f} ms  -> {B/((tf+tb)/K)*1e3:.0f} graphs/s")
