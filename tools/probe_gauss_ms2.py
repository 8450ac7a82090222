import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
def stats(tag):
    s = torch.cuda.memory_stats()
    print(tag, "reserved %.2f GB allocated %.2f GB retries %d cudaMalloc calls %d" % (s["reserved_bytes.all.current"] / 1e9, s["allocated_bytes.all.current"] / 1e9, s["num_alloc_retries"], s["segment.all.allocated"]))
if len(sys.argv) > 1:
    bench.hot_path_kernels(dev)
    stats("after kernels")
out = bench.sinkhorn_wallclock(dev)
stats("after wallclock")
g = torch.Generator().manual_seed(1)
x = torch.rand(1000000, 3, generator=g).to(dev); y = torch.rand(1000000, 3, generator=g).to(dev)
L = SamplesLoss("gaussian", blur=0.05, backend="multiscale")
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); L(x, y); torch.cuda.synchronize(); print("again", time.perf_counter() - t0)
stats("end")
