import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from geomloss_amd import hip
dev = torch.device("cuda:0")
n = 200_000
for D in (3, 4, 8, 16):
    g = torch.Generator().manual_seed(D)
    x, y = torch.rand(1, n, D, generator=g).to(dev), torch.rand(1, n, D, generator=g).to(dev)
    h = (torch.randn(1, n, generator=g) - math.log(n)).to(dev)
    v = (torch.rand(1, n, generator=g) / n).to(dev)
    t1 = bench.event_ms(lambda: hip.softmin_fwd_raw(x, y, h, 0.05, 1), 2)
    t2 = bench.event_ms(lambda: hip.kernel_conv_fwd_raw(hip.LAPLACIAN, x, y, v, 0.05), 2)
    t3 = bench.event_ms(lambda: hip.kernel_conv_fwd_raw(hip.ENERGY, x, y, v, 0.05), 2)
    p = float(n) * n
    print(f"D={D}: p=1 softmin {t1:.2f} ms {p/t1*1e3:.2e} pairs/s | laplacian {t2:.2f} ms {p/t2*1e3:.2e} | energy {t3:.2f} ms {p/t3*1e3:.2e}")
