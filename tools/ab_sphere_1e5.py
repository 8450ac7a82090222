"""GPU box: the reference protocol's sphere clouds, sinkhorn blur=.01 diameter=1, multiscale, N = 1e5 (and 1e4): loss + backward,
with the figures the dense switch sees.  Knobs: GEOMLOSS_HIP_DENSE_SWITCH, GEOMLOSS_HIP_SMALL_ROW_BLOCK."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss, sinkhorn_samples as ss
dev = torch.device("cuda:0")
seen = []
model = ss.dense_is_cheaper
ss.dense_is_cheaper = lambda kept, N, M, Cr, Cc: (seen.append((round(kept / (float(N) * M), 4), round(N / Cr, 1), model(kept, N, M, Cr, Cc))), seen[-1][2])[1]
goes = ss._goes_dense
ss._goes_dense = lambda *a: (lambda r: (seen.append(r), r)[1])(goes(*a))
for N in (10_000, 100_000):
    g = torch.Generator(device="cpu").manual_seed(N)
    x = torch.randn(N, 3, generator=g); x[:, 0] += 1; x = x / (2 * x.norm(dim=1, keepdim=True))
    y = torch.randn(N, 3, generator=g); y[:, 1] += 2; y = y / (2 * y.norm(dim=1, keepdim=True))
    a = torch.randn(N, generator=g).abs(); b = torch.randn(N, generator=g).abs()
    a, x, b, y = (a / a.sum()).to(dev), x.to(dev).requires_grad_(True), (b / b.sum()).to(dev), y.to(dev)
    for blur in (0.05, 0.01):
        loss = SamplesLoss("sinkhorn", p=2, blur=blur, diameter=1, backend="multiscale")
        ts = []
        for _ in range(6):
            del seen[:]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            L = loss(a, x, b, y); L.backward(); x.grad = None
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"N = {N} blur = {blur}: {sorted(ts)[len(ts) // 2]:.2f} ms  loss {L.item():.6e}  switch sees {seen}", flush=True)
