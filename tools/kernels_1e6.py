"""Launches every reduction of the hot path twice at N = M = 1e6 (for rocprofv3 kernel-trace / PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geomloss_amd import hip
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x, y, h, eps = bench.make_problem(n, dev, seed=7)
g = torch.randn(1, n, device=dev); v = torch.rand(1, n, device=dev) / n
out = hip.softmin_fwd_raw(x, y, h, eps, 2)
for _ in range(2):
    hip.softmin_fwd_raw(x, y, h, eps, 2)
    hip.softmin_fwd_raw(x, y, h, eps, 2, flags=hip.FLAG_F16X2)      # the headline launch of round 5 (xd_fwd_kernel<D = 3, f16 x 2>)
    hip.softmin_bwd_x_raw(x, y, h, out, g, eps, 2)
    hip.kernel_conv_fwd_raw(hip.GAUSSIAN, x, y, v, 0.05)
    hip.kernel_conv_bwd_x_raw(hip.GAUSSIAN, x, y, v, g, 0.05)
    hip.softmin_fwd_raw(x, y, h, 0.05, 1)
    hip.kernel_conv_fwd_raw(hip.LAPLACIAN, x, y, v, 0.05)
    hip.kernel_conv_fwd_raw(hip.ENERGY, x, y, v, 0.05)
torch.cuda.synchronize()
