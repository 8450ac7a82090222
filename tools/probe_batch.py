"""Per-rank work of the batch-sharded bench (BASELINE configs[3]): loss time for B = 256 / W problems on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
L = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)
for B in (256, 128, 64, 32):
    x, y = bench.cfg4_batch(dev, B, seed=2)
    for _ in range(3): L(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): L(x, y).sum()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"B={B:4d}: {t*1e3:7.3f} ms per loss  {bench.cfg4_pairs(B)/t:.3e} pairs/s  (ideal share of B=256: x{256//B})", flush=True)
