"""GPU box: throughput of the float64 soft-min / kernel product at a few sizes (pairs/s).  usage: bench_f64.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip
dev = torch.device("cuda:0")
for N, D in ((500, 3), (2197, 3), (20_000, 3), (200_000, 3), (200_000, 8)):
    g = torch.Generator().manual_seed(0)
    x, y = torch.rand(N, D, generator=g, dtype=torch.float64).to(dev), torch.rand(N, D, generator=g, dtype=torch.float64).to(dev)
    h = torch.zeros(N, dtype=torch.float64, device=dev)
    v = torch.full((N,), 1.0 / N, dtype=torch.float64, device=dev)
    for name, fn in (("softmin p=2", lambda: hip.softmin(0.01, x, y, h)), ("softmin p=1", lambda: hip.softmin(0.1, x, y, h, p=1)),
                     ("gaussian", lambda: hip.kernel_conv("gaussian", x, y, v, 0.1))):
        for _ in range(2):
            fn()
        reps = 20 if N <= 20_000 else 3
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(f"N = M = {N:7d} D = {D}: {name:12s} {dt * 1e3:9.3f} ms  {N * N / dt:.3e} pairs/s", flush=True)
