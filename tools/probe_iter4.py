import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
def t(L, *a, n=6):
    for _ in range(2): L(*a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): L(*a)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
x, y = bench.cfg4_batch(dev, 256, seed=2)
print("B=256 bf16 4096:", "%.3f ms" % t(SamplesLoss("sinkhorn", backend="online", **bench.CFG4), x, y), flush=True)
g = torch.Generator().manual_seed(0)
for N in (30000, 50000, 70000, 100000):
    x, y = torch.rand(N, 3, generator=g).to(dev), torch.rand(N, 3, generator=g).to(dev)
    print(f"N={N}:", "%.3f ms" % t(SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online"), x, y, n=4), flush=True)
