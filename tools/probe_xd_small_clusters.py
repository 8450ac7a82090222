"""GPU box: the two-scale loss on D = 4 clouds with the reference's ~2000-cluster rule at small N (clusters of 5-50 points): what gathered
tiles buy the block-sparse D > 3 soft-min.  GEOMLOSS_HIP_LIB selects another build for an A/B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
for n in (10_000, 30_000, 100_000):
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(n, 4, generator=g).to(dev), torch.rand(n, 4, generator=g).to(dev)
    loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale")
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L = loss(x, y)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"N = {n}: {dt * 1e3:.2f} ms  loss {L.item():.6e}", flush=True)
