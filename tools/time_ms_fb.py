"""GPU box: median time of the two-scale loss, forward and forward + backward.  usage: time_ms_fb.py [N] [REPS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
g = torch.Generator().manual_seed(1)
x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale")
def med(fn):
    ts = []
    for _ in range(reps + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts[2:])[reps // 2]
def fb():
    xg = x.clone().requires_grad_(True)
    torch.autograd.grad(L(xg, y), [xg])
print(f"N = {n}: fwd {med(lambda: L(x, y)):.2f} ms   fwd+bwd {med(fb):.2f} ms   env " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith(("GLHIP_", "GEOMLOSS_"))), flush=True)
