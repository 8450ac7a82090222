"""GPU box: kernel norms at N = M = 1e6 on the union cloud (kernel_samples._kernel_loss_union): seconds per call, value only
and with a gradient, for the three kernels, on bench.py's clouds (seed 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss

dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
g = torch.Generator().manual_seed(1)
x = torch.rand(n, 3, generator=g).to(dev)
y = torch.rand(n, 3, generator=g).to(dev)
for name in ("gaussian", "energy", "laplacian"):
    loss = SamplesLoss(name, blur=0.05, backend="online")
    for mode in ("value", "grad", "potentials"):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "value":
                with torch.no_grad():
                    L = loss(x, y)
            elif mode == "grad":
                xg = x.clone().requires_grad_(True)
                L = loss(xg, y)
                torch.autograd.grad(L, [xg])
            else:
                L = SamplesLoss(name, blur=0.05, backend="online", potentials=True)(x, y)[0].sum()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"{name:10s} {mode:10s} N=M={n}: {min(ts[1:]):.4f} s (first {ts[0]:.3f})  value {float(L):.9e}", flush=True)
