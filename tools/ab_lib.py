"""A/B two builds of the library on one soft-min size: python tools/ab_lib.py <lib.so> <N>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip
import ctypes
_probe = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name in list(hip.SIGNATURES):
    if not hasattr(_probe, name):
        hip.SIGNATURES.pop(name)
hip.load_library(os.path.abspath(sys.argv[1]))
N = int(sys.argv[2]); dev = torch.device("cuda:0"); torch.manual_seed(0)
x, y = torch.rand(1, N, 3, device=dev), torch.rand(1, N, 3, device=dev)
h = torch.full((1, N), -float(torch.log(torch.tensor(float(N)))), device=dev)
for _ in range(3): hip.softmin_fwd_raw(x, y, h, 0.0025, 2)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); hip.softmin_fwd_raw(x, y, h, 0.0025, 2); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(os.path.basename(sys.argv[1]), N, "min %.3f ms" % min(ts), "ws bytes", hip.load_library().glhip_workspace_bytes(1, N, N, 3, 0))
