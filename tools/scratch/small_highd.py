"""GPU box: online losses on a few thousand points in D = 4 / 8 (p = 2) and D = 3 / 5 (p = 1); median of 60 calls.
Run with GLHIP_TINY_MULTI_PAIRS=0 for the split rule of round 4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
out = []
for n, D, p in ((1000, 4, 2), (2000, 4, 2), (2000, 8, 2), (2000, 16, 2), (1000, 3, 1), (2000, 3, 1), (2000, 5, 1)):
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(n, D, generator=g).to(dev), torch.rand(n, D, generator=g).to(dev)
    loss = SamplesLoss("sinkhorn", p=p, blur=0.05, backend="online")
    ts = []
    for r in range(70):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L = loss(x, y)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[10:])
    out.append(f"N={n} D={D} p={p}: {ts[len(ts) // 2] * 1e3:.3f} ms")
print(f"tiny={os.environ.get('GLHIP_TINY_MULTI_PAIRS', 'default')}: " + " | ".join(out), flush=True)
