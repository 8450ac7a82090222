"""A/B of the gradient kernels at N = M = n (default 1e6): 16x16x32 (default for D <= 3) vs the transposed 32x32x16 kernel
(GLHIP_FLAG_T32, csrc/glhip_wsum_t32.h), and the D > 3 gradients.   usage: python tools/ab_grad_kernels.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip
from bench import event_ms, make_problem

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
x, y, h, eps = make_problem(n, dev, seed=7)
g = torch.randn(1, n, device=dev)
v = torch.rand(1, n, device=dev) / n
out = hip.softmin_fwd_raw(x, y, h, eps, 2)
for name, fl in (("16x16x32", 0),):
    t1 = event_ms(lambda: hip.softmin_bwd_x_raw(x, y, h, out, g, eps, 2, flags=fl), 3)
    t2 = event_ms(lambda: hip.kernel_conv_bwd_x_raw(hip.GAUSSIAN, x, y, v, g, 0.05, flags=fl), 3)
    t3 = event_ms(lambda: hip.kernel_conv_fwd_grad_raw(hip.GAUSSIAN, x, y, v, 0.05, flags=fl), 3)
    print(f"D=3 {name:22s}: soft-min gradient {t1:7.2f} ms   gaussian gradient {t2:7.2f} ms   product + gradient {t3:7.2f} ms", flush=True)
gd = torch.Generator().manual_seed(11)
for D in (4, 8, 16):
    xd, yd = torch.rand(1, n, D, generator=gd).to(dev), torch.rand(1, n, D, generator=gd).to(dev)
    od = hip.softmin_fwd_raw(xd, yd, h, eps, 2)
    t1 = event_ms(lambda: hip.softmin_bwd_x_raw(xd, yd, h, od, g, eps, 2), 2)
    t3 = event_ms(lambda: hip.kernel_conv_fwd_grad_raw(hip.GAUSSIAN, xd, yd, v, 0.1), 2)
    print(f"D={D}: soft-min gradient {t1:7.2f} ms   gaussian product + gradient {t3:7.2f} ms", flush=True)
