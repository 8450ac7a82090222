"""The headline soft-min (N = M = 1e6, 3-D, eps = .05^2) timed on three dual vectors h = log b + g / eps: the synthetic one of
bench.py (log-weights + noise of 4 exponent units), and g = the converged dual potential g_ab of
SamplesLoss("sinkhorn", backend="multiscale", debias=False, potentials=True) on two samples of one law and on shifted clouds
(y = 0.6 y + 0.3: g spans tens of exponent units).  The kernel takes its maximum lazily (a tile whose partial sum overflows the
running scale is redone): this is the check that the redo path does not show with real potentials.  ms per launch, HIP events."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss, hip
import bench
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
x, y, h, eps = bench.make_problem(n, dev, 0)
loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale", debias=False, potentials=True)
duals = [("synthetic (bench.py)", y, h)]
for tag, yy in (("converged, two samples of one law", y[0]), ("converged, shifted clouds", (0.6 * y[0] + 0.3).contiguous())):
    F, G = loss(x[0], yy)
    duals.append((tag, yy[None].contiguous(), (torch.full((n,), -math.log(n), device=dev) + G.view(-1) / eps)[None].contiguous()))
for tag, yy, hh in duals:
    ms = bench.event_ms(lambda: hip.softmin_fwd_raw(x, yy, hh, eps, 2), reps=5)
    spread = (hh.max() - hh.min()).item()
    print(f"{tag:36s} {ms:8.3f} ms   {n * n / ms * 1e-9:6.2f}e12 pairs/s   range of h: {spread:6.1f}", flush=True)
