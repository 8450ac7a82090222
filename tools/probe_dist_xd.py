"""Distance reductions for 4 <= D <= 16 on the matrix cores (csrc/glhip_dist_xd.h) against the float64 oracle and against the
one-thread-per-row kernel of glhip_generic.h (GLHIP_FLAG_NO_MFMA), with timings at 2e5 x 2e5."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from geomloss_amd import hip
from oracle import oracle_torch64 as o64

dev = torch.device("cuda:0")


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


for D in (4, 5, 8, 16):
    g = torch.Generator().manual_seed(D)
    N, M = 30_000, 40_000
    x, y = torch.rand(N, D, generator=g).to(dev), (torch.rand(M, D, generator=g) * 0.8 + 0.1).to(dev)
    y[:2000] = x[:2000]                    # coincident and ...
    y[2000:4000] = x[2000:4000] + 1e-4     # ... near pairs: the exact path
    h = (torch.randn(M, generator=g) * 2 - math.log(M)).to(dev)
    v = (torch.rand(M, generator=g) / M).to(dev)
    eps = 0.05 * math.sqrt(D / 3.0)
    line = [f"D={D:2d}"]
    refs = (o64.softmin(eps, x, y, h, p=1, device=dev), o64.kconv("laplacian", x, y, v, eps, device=dev), o64.kconv("energy", x, y, v, eps, device=dev))
    for name, fl in (("mfma", 0), ("generic", hip.FLAG_NO_MFMA)):
        f = hip.softmin(eps, x, y, h, p=1, flags=fl).cpu().numpy()
        kl = hip.kernel_conv("laplacian", x, y, v, eps, flags=fl).cpu().numpy()
        ke = hip.kernel_conv("energy", x, y, v, eps, flags=fl).cpu().numpy()
        line.append(f"{name}: softmin abs {np.abs(f - refs[0]).max():.2e} laplacian rel {relerr(kl, refs[1]):.2e} energy rel {relerr(ke, refs[2]):.2e}")
    print(" | ".join(line), flush=True)
    n = 200_000
    xb, yb = torch.rand(1, n, D, generator=g).to(dev), torch.rand(1, n, D, generator=g).to(dev)
    hb, vb = (torch.randn(1, n, generator=g) - math.log(n)).to(dev), (torch.rand(1, n, generator=g) / n).to(dev)
    p = float(n) * n
    t = {}
    for name, fl in (("mfma", 0), ("generic", hip.FLAG_NO_MFMA)):
        t[name] = (bench.event_ms(lambda: hip.softmin_fwd_raw(xb, yb, hb, eps, 1, flags=fl), 2),
                   bench.event_ms(lambda: hip.kernel_conv_fwd_raw(hip.LAPLACIAN, xb, yb, vb, eps, flags=fl), 2),
                   bench.event_ms(lambda: hip.kernel_conv_fwd_raw(hip.ENERGY, xb, yb, vb, eps, flags=fl), 2))
    print("      2e5 x 2e5: " + " | ".join(f"{nm} {t['generic'][k]:6.2f} -> {t['mfma'][k]:6.2f} ms ({p / t['mfma'][k] * 1e3:.2e} pairs/s)"
                                            for k, nm in enumerate(("softmin p=1", "laplacian", "energy"))), flush=True)
