"""GPU box: two launches each of the D = 4, 8, 12, 16 soft-min forward and the D = 8 gaussian product at N = M = 1e6 (glhip_softmin_xd.h),
the workload rocprofv3 is pointed at by tools/profile_kernels.sh <tag> kernels_xd_1e6.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip

dev = torch.device("cuda:0")
n = 1_000_000
for D in (4, 8, 12, 16):
    g = torch.Generator().manual_seed(D)
    x = torch.rand(1, n, D, generator=g).to(dev)
    y = torch.rand(1, n, D, generator=g).to(dev)
    h = (torch.randn(1, n, generator=g) * 2).to(dev)
    v = (torch.rand(1, n, generator=g) / n).to(dev)
    for _ in range(2):
        hip.softmin_fwd_raw(x, y, h, 0.05 ** 2 * D / 3, 2)
    if D == 8:
        for _ in range(2):
            hip.kernel_conv_fwd_raw(hip.GAUSSIAN, x, y, v, 0.05 * (D / 3) ** 0.5)
torch.cuda.synchronize()
