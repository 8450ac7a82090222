import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
from geomloss_amd import hip
from oracle import oracle_torch64 as o64
dev = torch.device("cuda:0")
for D in (1, 2, 3):
    for n in (2000, 10_000, 30_000, 100_000):
        g = torch.Generator().manual_seed(D + n)
        x, y = torch.rand(1, n, D, generator=g).to(dev), torch.rand(1, n, D, generator=g).to(dev)
        h = (torch.randn(1, n, generator=g) - math.log(n)).to(dev)
        t = bench.event_ms(lambda: hip.softmin_fwd_raw(x, y, h, 0.05, 1), 5)
        out = hip.softmin_fwd_raw(x, y, h, 0.05, 1)[0].cpu().numpy()
        ref = o64.softmin(0.05, x[0], y[0], h[0], p=1, device=dev)
        print(f"D={D} N={n}: {t*1e3:8.1f} us  {float(n)*n/t*1e3:.2e} pairs/s  abs err {np.abs(out-ref).max():.2e}", flush=True)
