import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geomloss_amd import hip
from oracle import oracle_torch64 as o64
cuda = torch.device("cuda:0")
N, M = 70_000, 8_000
g = torch.Generator().manual_seed(2)
x, y = torch.rand(N, 3, generator=g).to(cuda), torch.rand(M, 3, generator=g).to(cuda)
h = (torch.randn(M, generator=g) * 2).to(cuda)
eps = 0.005
ref = o64.softmin(eps, x, y, h, p=1, device=cuda)
out = hip.softmin(eps, x, y, h, p=1).cpu().numpy()
err = np.abs(out - ref)
worst = np.argsort(-err)[:8]
plan = hip._plan_cache[-1][3]
perm = plan.perm.long().cpu().numpy(); inv = np.empty_like(perm); inv[perm] = np.arange(N)
rng = plan.ranges.ranges_i.cpu().numpy()
d = torch.cdist(x[torch.from_numpy(worst).to(cuda)].double(), y.double())
print("clusters", rng.shape[0], "rows/cluster median", int(np.median(rng[:,1]-rng[:,0])), "max", int((rng[:,1]-rng[:,0]).max()))
for k, i in enumerate(worst):
    p = inv[i]; c = np.searchsorted(rng[:, 1], p, side="right")
    dm, jm = d[k].min(0)
    # dominant column of the soft-min
    u = h.double() - d[k] / eps
    jd = int(u.argmax())
    print(f"row {i}: err {err[i]:.2e} f {ref[i]:.4f} cluster {c} size {rng[c,1]-rng[c,0]} nearest col d={dm.item():.2e}; dominant col d={d[k][jd].item():.3e} u-gap to 2nd {float(torch.sort(u, descending=True)[0][0] - torch.sort(u, descending=True)[0][1]):.2f}")
print("errors percentiles", np.percentile(err, [50, 90, 99, 99.9, 100]))
