"""Reference-protocol gaussian config (sphere clouds, blur=.1, truncate=3) at N: timings + block statistics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss, cluster
import geomloss_amd.kernel_samples as ks
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = 3
g = torch.Generator(device="cpu").manual_seed(N)
x = torch.randn(N, D, generator=g); x[:, 0] += 1; x = x / (2 * x.norm(dim=1, keepdim=True))
y = torch.randn(N, D, generator=g); y[:, 1] += 2; y = y / (2 * y.norm(dim=1, keepdim=True))
a = torch.randn(N, generator=g).abs(); b = torch.randn(N, generator=g).abs()
a, x, b, y = (a / a.sum()).to(dev), x.to(dev).requires_grad_(True), (b / b.sum()).to(dev), y.to(dev)
orig = cluster.from_matrix
def fm(ri, rj, keep):
    out = orig(ri, rj, keep)
    ni, nj = (ri[:, 1] - ri[:, 0]).double(), (rj[:, 1] - rj[:, 0]).double()
    kept = float((ni[:, None] * nj[None, :] * keep).sum() / (ni.sum() * nj.sum()))
    print(f"  mask {tuple(keep.shape)}: kept clusters {float(keep.double().mean()):.3f}, kept pairs {kept:.3f}, rows/cluster min {int(ni.min())} med {int(ni.median())} max {int(ni.max())}, "
          f"intervals {out.redranges_j.shape[0]}")
    return out
ks.from_matrix = fm
for name, kw in (("multiscale", dict(truncate=3)), ("online", dict())):
    L = SamplesLoss("gaussian", blur=0.1, backend=name, verbose=True, **kw)
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v = L(a, x, b, y); torch.cuda.synchronize(); t1 = time.perf_counter()
        v.backward(); x.grad = None; torch.cuda.synchronize(); t2 = time.perf_counter()
        if r == 0:
            ks.from_matrix = orig
        print(f"{name} N={N} rep {r}: fwd {t1 - t0:.4f} bwd {t2 - t1:.4f} loss {v.item():.6e}", flush=True)
