"""GPU box: the two-scale Sinkhorn loss of BASELINE configs[2] (N = M = 1e6, 3-D, same law, seed 1: the clouds bench.py times),
forward + backward, three times — the workload of
    MIN_NS=5e6 PAIRS=2.1e11 tools/profile_kernels.sh <tag> multiscale_1e6.py
whose summary lists the fine-level block-sparse kernels (soft-min forward: 12 launches per loss, value + gradient: 2) with their
counters.  PAIRS = kept pairs of one fine-level reduction (21 % of 1e12)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator().manual_seed(1)
x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale")
for _ in range(reps):
    xg = x.clone().requires_grad_(True)
    val = L(xg, y)
    (gx,) = torch.autograd.grad(val, [xg])
torch.cuda.synchronize()
print(val.item(), gx.abs().max().item())
