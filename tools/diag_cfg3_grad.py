"""GPU box: where the dL/dx error of the 1e6 same-law two-scale loss sits (seed from argv): quantiles, the worst rows, their surroundings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geomloss_amd import SamplesLoss
from oracle import oracle_torch64 as o64
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
g = torch.Generator().manual_seed(seed)
x, y = torch.rand(N, 3, generator=g).to(dev), torch.rand(N, 3, generator=g).to(dev)
kw = dict(p=2, blur=0.05)
xg = x.clone().requires_grad_(True)
L = SamplesLoss("sinkhorn", backend="multiscale", **kw)(xg, y)
(gx,) = torch.autograd.grad(L, [xg])
gx = gx.cpu().numpy().astype(np.float64) * N
torch.cuda.empty_cache()
a = np.full(N, 1.0 / N)
ref = o64.sinkhorn_multiscale(a, x, a, y, full=True, device=dev, **kw)
rg = ref["gx"] * N
err = np.abs(gx - rg).max(1)
print(f"seed {seed}: max |ref| {np.abs(rg).max():.3e}  rms |ref| {np.sqrt((rg**2).sum(1).mean()):.3e}  max err {err.max():.3e}  rel {err.max() / np.abs(rg).max():.3e}")
print("error quantiles 50/90/99/99.9/99.99/max:", " ".join(f"{np.quantile(err, q):.2e}" for q in (0.5, 0.9, 0.99, 0.999, 0.9999, 1.0)))
xs = x.cpu().numpy()
for i in np.argsort(-err)[:8]:
    d = np.sqrt(((xs - xs[i]) ** 2).sum(1)); d[i] = 9
    dy = np.sqrt(((y.cpu().numpy() - xs[i]) ** 2).sum(1))
    print(f"row {i}: err {err[i]:.3e} ref {rg[i]} got {gx[i]} x {xs[i]}  nearest x {d.min():.2e} nearest y {dy.min():.2e}  voxel frac {np.modf(xs[i] / (1 / 2000 ** (1 / 3)) )[0]}")
w = np.argmax(np.abs(rg).max(1))
print("row of max |ref|:", w, rg[w], gx[w], xs[w])
