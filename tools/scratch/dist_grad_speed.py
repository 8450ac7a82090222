import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from geomloss_amd import hip
dev = torch.device("cuda:0")
n = 200_000
for D in (4, 8, 16):
    g = torch.Generator().manual_seed(D)
    x, y = torch.rand(1, n, D, generator=g).to(dev), torch.rand(1, n, D, generator=g).to(dev)
    h = (torch.randn(1, n, generator=g) - math.log(n)).to(dev)
    v = (torch.rand(1, n, generator=g) / n).to(dev)
    gg = torch.randn(1, n, generator=g).to(dev)
    out = hip.softmin_fwd_raw(x, y, h, 0.1, 1)
    p = float(n) * n
    line = [f"D={D}"]
    for name, fl in (("generic", hip.FLAG_NO_MFMA), ("mfma", 0)):
        t1 = bench.event_ms(lambda: hip.softmin_bwd_x_raw(x, y, h, out, gg, 0.1, 1, flags=fl), 2)
        t2 = bench.event_ms(lambda: hip.kernel_conv_bwd_x_raw(hip.LAPLACIAN, x, y, v, gg, 0.1, flags=fl), 2)
        t3 = bench.event_ms(lambda: hip.kernel_conv_bwd_x_raw(hip.ENERGY, x, y, v, gg, 0.1, flags=fl), 2)
        line.append(f"{name}: softmin p=1 grad {t1:.1f} ms ({p/t1*1e3:.2e}/s) laplacian grad {t2:.1f} ms ({p/t2*1e3:.2e}/s) energy grad {t3:.1f} ms ({p/t3*1e3:.2e}/s)")
    print(" | ".join(line), flush=True)
