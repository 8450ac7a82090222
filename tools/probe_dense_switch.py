"""GPU box: the two-scale loss with its truncated fine level block-sparse ("0"), dense ("always") and chosen by the cost model ("1",
geomloss_amd/sinkhorn_samples.py: dense_is_cheaper), with the figures the model sees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss, sinkhorn_samples as ss
dev = torch.device("cuda:0")
seen = []
model = ss.dense_is_cheaper


def spy(kept, N, M, Cr, Cc):
    seen.append((kept / (float(N) * M), N / Cr, M / Cc, model(kept, N, M, Cr, Cc)))
    return seen[-1][3]


def labels4d(t, scale=0.08):
    q = (t[:, :3] / scale).floor().long()
    return torch.unique((q[:, 0] * 64 + q[:, 1]) * 64 + q[:, 2], return_inverse=True)[1].int()


cases = [(3, 10_000), (3, 30_000), (3, 100_000), (3, 300_000), (3, 1_000_000), (2, 10_000), (2, 100_000), (4, 10_000), (4, 100_000),
         (8, 10_000), ("4 labels", 200_000)]
for D, n in cases:
    g = torch.Generator().manual_seed(3)
    d = 4 if D == "4 labels" else D
    x, y = torch.rand(n, d, generator=g).to(dev), torch.rand(n, d, generator=g).to(dev)
    w = torch.full((n,), 1.0 / n, device=dev)
    args = (labels4d(x), w, x, labels4d(y), w, y) if D == "4 labels" else (x, y)
    loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale")
    out = []
    for mode in ("0", "always", "1"):
        ss.set_dense_switch(mode)
        ss.dense_is_cheaper = spy if mode == "1" else model
        del seen[:]
        for r in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            L = loss(*args)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out.append((dt, L.item()))
    k = seen[0] if seen else (float("nan"),) * 4
    print(f"D = {D} N = {n}: sparse {out[0][0] * 1e3:8.2f} ms  dense {out[1][0] * 1e3:8.2f} ms  model {out[2][0] * 1e3:8.2f} ms"
          f"   kept {k[0]:.3f} cluster {k[1]:.0f} x {k[2]:.0f} -> dense={k[3]}   loss {out[0][1]:.7e} / {out[1][1]:.7e} "
          f"(rel {abs(out[0][1] - out[1][1]) / abs(out[0][1]):.1e})", flush=True)
