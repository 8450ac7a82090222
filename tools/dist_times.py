"""Public-entry-point timings of the distance-type reductions at N = M = 1e6 (two voxel sorts included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip
from bench import event_ms, make_problem
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
x, y, h, eps = make_problem(n, dev, seed=7)
v = torch.rand(n, device=dev) / n
print("soft-min p=1   %.2f ms" % event_ms(lambda: hip.softmin(0.05, x[0], y[0], h[0], p=1), 2))
print("laplacian      %.2f ms" % event_ms(lambda: hip.kernel_conv("laplacian", x[0], y[0], v, 0.05), 2))
print("energy         %.2f ms" % event_ms(lambda: hip.kernel_conv("energy", x[0], y[0], v, 0.05), 2))
