import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
L = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)
for B in (256, 32):
    x, y = bench.cfg4_batch(dev, B, seed=2)
    ts = []
    for _ in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v = L(x, y)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(B, " ".join(f"{t:.1f}" for t in ts))
    s = torch.cuda.memory_stats()
    print("   reserved %.2f GB, cudaMalloc segments %d, alloc retries %d" % (s["reserved_bytes.all.current"] / 1e9, s["segment.all.allocated"], s["num_alloc_retries"]))
from torch.profiler import profile, ProfilerActivity
x, y = bench.cfg4_batch(dev, 32, seed=2)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    L(x, y); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=60))
