"""f16 x 2 exponents (GLHIP_FLAG_F16X2) against the default bf16 x 3 layout: accuracy against the float64 oracle at 4e4 x 5e4, time at 1e6.

    python tools/probe_f16x2.py [--dims 3,4,5,8,12,16] [--no-time]
"""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from geomloss_amd import hip
from oracle import oracle_torch64 as o64

ap = argparse.ArgumentParser()
ap.add_argument("--dims", default="3,4,5,8,12,16")
ap.add_argument("--no-time", action="store_true")
ap.add_argument("--points", type=int, default=1_000_000)
args = ap.parse_args()
dev = torch.device("cuda:0")
H2 = hip.FLAG_F16X2


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


for D in [int(d) for d in args.dims.split(",")]:
    g = torch.Generator().manual_seed(100 + D)
    N, M = 40_000, 50_000
    x, y = torch.rand(N, D, generator=g).to(dev), (torch.rand(M, D, generator=g) * 0.8 + 0.1).to(dev)
    eps, blur = 0.05**2, 0.05 * math.sqrt(D / 3.0)
    h = (torch.randn(M, generator=g) * 2 - math.log(M)).to(dev)
    h[::97] = -float("inf")
    gg = torch.randn(N, generator=g).to(dev)
    v = (torch.rand(M, generator=g) / M).to(dev)
    f_ref = o64.softmin(eps, x, y, h, device=dev)
    g_ref = o64.softmin_grad_x(eps, x, y, h, gg, device=dev)
    k_ref = o64.kconv("gaussian", x, y, v, blur, device=dev)
    line = [f"D={D:2d}"]
    for name, fl in (("bf16x3", 0), ("f16x2", H2)):
        xt = x.clone().requires_grad_(True)
        out = hip.softmin(eps, xt, y, h, flags=fl)
        (gx,) = torch.autograd.grad(out, [xt], grad_outputs=gg)
        k = hip.kernel_conv("gaussian", x, y, v, blur, flags=fl)
        line.append(f"{name}: softmin abs {np.abs(out.detach().cpu().numpy() - f_ref).max():.2e} grad rel {relerr(gx.cpu().numpy(), g_ref):.2e} "
                    f"gauss rel {relerr(k.cpu().numpy(), k_ref):.2e}")
    print(" | ".join(line), flush=True)
    if args.no_time:
        continue
    n = args.points
    gd = torch.Generator().manual_seed(11)
    xd, yd = torch.rand(1, n, D, generator=gd).to(dev), torch.rand(1, n, D, generator=gd).to(dev)
    _, _, hh, _ = bench.make_problem(n, dev, seed=7)
    gN = torch.randn(1, n, device=dev)
    vN = torch.rand(1, n, device=dev) / n
    out = hip.softmin_fwd_raw(xd, yd, hh, eps, 2)
    t = {}
    for name, fl in (("bf16x3", 0), ("f16x2", H2)):
        t[name] = (bench.event_ms(lambda: hip.softmin_fwd_raw(xd, yd, hh, eps, 2, flags=fl), 2),
                   bench.event_ms(lambda: hip.softmin_bwd_x_raw(xd, yd, hh, out, gN, eps, 2, flags=fl), 1),
                   bench.event_ms(lambda: hip.kernel_conv_fwd_raw(hip.GAUSSIAN, xd, yd, vN, 2 * blur, flags=fl), 1))
    print(f"      1e6 x 1e6: forward {t['bf16x3'][0]:7.2f} -> {t['f16x2'][0]:7.2f} ms | gradient {t['bf16x3'][1]:7.2f} -> {t['f16x2'][1]:7.2f} ms | "
          f"gaussian product {t['bf16x3'][2]:7.2f} -> {t['f16x2'][2]:7.2f} ms", flush=True)
