"""GPU box: single launches of the D > 3 distance kernels on small clouds (energy / laplacian products, p = 1 soft-min): us per call.
GLHIP_DIST_MULTI_MIN_COLS=0 for the split rule alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geomloss_amd import hip
dev = torch.device("cuda:0")
out = []
for n, D in ((1000, 8), (2000, 8), (4000, 5), (8000, 8)):
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(n, D, generator=g).to(dev), torch.rand(n, D, generator=g).to(dev)
    v = torch.full((n,), 1.0 / n, device=dev)
    h = torch.zeros(n, device=dev)
    for name, fn in (("energy", lambda: hip.kernel_conv("energy", x, y, v, 0.1)), ("p1", lambda: hip.softmin(0.1, x, y, h, p=1))):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(50):
            fn()
        b.record(); torch.cuda.synchronize()
        out.append(f"N={n} D={D} {name}: {a.elapsed_time(b) / 50 * 1e3:.1f} us")
print(f"min_cols={os.environ.get('GLHIP_DIST_MULTI_MIN_COLS', 'default')}: " + " | ".join(out), flush=True)
