"""The distance-type reductions at N = M = 1e6 through the public entry points (two voxel sorts + the matrix-core distance
kernel), twice each — for tools/profile_kernels.sh <tag> dist_1e6.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from geomloss_amd import hip
dev = torch.device("cuda:0")
x, y, h, eps = bench.make_problem(1_000_000, dev, seed=7)
v = torch.rand(1_000_000, device=dev) / 1e6
for _ in range(2):
    hip.kernel_conv("energy", x[0], y[0], v, 0.05)
    hip.kernel_conv("laplacian", x[0], y[0], v, 0.05)
    hip.softmin(0.05, x[0], y[0], h[0], p=1)
torch.cuda.synchronize()
