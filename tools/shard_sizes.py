"""What each rank of the batch-sharded bench (BASELINE configs[3]) has to do at 1 / 2 / 4 / 8 GPUs, timed on ONE GPU:
the loss of B/W problems of 4096 x 4096 points, W = 1, 2, 4, 8 — ms per loss and the strong-scaling efficiency it implies
(the scalar all-reduce aside).  usage: python tools/shard_sizes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
import bench
dev = torch.device("cuda:0")
loss = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)
X, Y = bench.cfg4_batch(torch.device("cpu"), 256, seed=2)
base = None
for W in (1, 2, 4, 8):
    x, y = X[:256 // W].to(dev), Y[:256 // W].to(dev)
    for _ in range(3): loss(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 10
    for _ in range(reps): L = loss(x, y)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    base = base or ms
    print(f"W={W}  B/W={256 // W:3d}   {ms:7.3f} ms/loss   implied strong-scaling efficiency {base / (W * ms):.3f}", flush=True)
