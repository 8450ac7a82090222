"""SamplesLoss kernel norms at N = M = 1e6, forward and forward + backward (seconds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
g = torch.Generator().manual_seed(1)
x0, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
for name, kw in (("energy", {}), ("laplacian", dict(blur=0.05)), ("gaussian", dict(blur=0.05))):
    for backward in (False, True):
        x = x0.clone().requires_grad_(backward)
        loss = SamplesLoss(name, backend="online", **kw)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            L = loss(x, y)
            if backward:
                torch.autograd.grad(L, [x])
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{name:10s} {'fwd+bwd' if backward else 'fwd    '}: {min(ts[1:]):.4f} s   loss {L.item():.6e}", flush=True)
