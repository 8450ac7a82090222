import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geomloss_amd import SamplesLoss, hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for D, n, be in ((4, 10_000, "online"), (8, 10_000, "online"), (4, 2_000, "online"), (3, 10_000, "online"), (4, 30_000, "online"), (5, 10_000, "online")):
    for p in (2, 1):
        x, y = torch.rand(n, D, generator=g).to(dev), torch.rand(n, D, generator=g).to(dev)
        L = SamplesLoss("sinkhorn", p=p, blur=0.05 if p == 2 else 0.1, backend=be)
        for _ in range(3): v = L(x, y)
        hip.settle_host(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): v = L(x, y)
        torch.cuda.synchronize()
        print(f"D={D} N={n} p={p} {be}: {(time.perf_counter()-t0)/20*1e3:.3f} ms  loss {v.item():.6e}", flush=True)
