"""GPU box: three two-scale losses on uniform clouds, for kernel traces.  usage: run_ms.py N D"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
n, D = int(float(sys.argv[1])), int(sys.argv[2])
g = torch.Generator().manual_seed(3)
x, y = torch.rand(n, D, generator=g).to(dev), torch.rand(n, D, generator=g).to(dev)
loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale", verbose=len(sys.argv) > 3)
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L = loss(x, y)
    torch.cuda.synchronize()
    print(f"rep {r}: {(time.perf_counter() - t0) * 1e3:.2f} ms  loss {L.item():.6e}", flush=True)
