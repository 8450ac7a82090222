"""Runs one BASELINE config end to end (for rocprofv3 --kernel-trace --stats and wall-clock checks).
usage: python tools/run_config.py {multiscale|online|batched|gaussian|gaussian_ms|energy} [reps]   (VERBOSE=1: verbose=True)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss

which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
VERBOSE = os.environ.get("VERBOSE", "0") == "1"     # verbose=True diverts kernel_truncation to the torch fallback (it prints kept fractions)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
if which == "multiscale":
    n = 1_000_000
    x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
    loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale", verbose=VERBOSE)
elif which == "online":
    n = 100_000
    x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
    loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online")
elif which == "batched":
    B = int(os.environ.get("B", 256))
    x = torch.rand(B, 4096, 3, generator=g).to(dev).bfloat16()
    y = torch.rand(B, 4096, 3, generator=g).to(dev).bfloat16()
    loss = SamplesLoss("sinkhorn", p=2, blur=0.05, diameter=1.8, backend="online")
elif which == "gaussian":
    n = 1_000_000
    x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
    loss = SamplesLoss("gaussian", blur=0.05, backend="online")
elif which == "energy":
    n = 1_000_000
    x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
    loss = SamplesLoss("energy", backend="online")
elif which == "gaussian_ms":
    n = 1_000_000
    x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
    loss = SamplesLoss("gaussian", blur=float(os.environ.get("BLUR", 0.05)), truncate=float(os.environ.get("TRUNC", 5)),
                       backend="multiscale", verbose=VERBOSE)
x.requires_grad_(True)
for r in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L = loss(x, y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (gx,) = torch.autograd.grad(L.sum(), [x])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{which} rep {r}: fwd {t1-t0:.4f} s  bwd {t2-t1:.4f} s  loss {L.sum().item():.6e}", flush=True)
