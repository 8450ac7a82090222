"""Per-call distribution of the batch-sharded workload (BASELINE configs[3]) at the shard sizes of 1 / 2 / 4 / 8 GPUs.

    python tools/probe_shards.py [--calls 200] [--sizes 256,128,64,32] [--reverse] [--no-fusion]

One call = one forward ``SamplesLoss('sinkhorn', online)(x, y).sum()`` on B problems of 4096 x 4096 bf16 points.  Every call is
bracketed by HIP events on the launch stream AND by host clocks (time to queue the call, without a synchronisation), so that a
slow mean can be told apart: a uniform slowdown (the launch plan), a few long stalls (allocator, host), or a host-bound loop
(queueing time ~ call time).  Prints median / p90 / p99 / max, the stall count (> 1.3 x median) and the whole-run wall time per call.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from geomloss_amd import SamplesLoss, sinkhorn_samples


def pct(v, q):
    s = sorted(v)
    return s[min(len(s) - 1, int(q * len(s)))]


def probe(B, calls, dev, warm=5):
    L = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)
    x, y = bench.cfg4_batch(dev, B, seed=2)
    for _ in range(warm):
        L(x, y).sum()
    torch.cuda.synchronize()
    a = [torch.cuda.Event(enable_timing=True) for _ in range(calls)]
    b = [torch.cuda.Event(enable_timing=True) for _ in range(calls)]
    host = []
    t0 = time.perf_counter()
    for k in range(calls):
        a[k].record()
        h0 = time.perf_counter()
        L(x, y).sum()
        host.append((time.perf_counter() - h0) * 1e3)
        b[k].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / calls * 1e3
    gpu = [s.elapsed_time(e) for s, e in zip(a, b)]
    med = pct(gpu, 0.5)
    stalls = sum(1 for t in gpu if t > 1.3 * med)
    pairs = bench.cfg4_pairs(B)
    print(f"B={B:4d}: wall {wall:7.3f} ms/call | events: median {med:7.3f}  p90 {pct(gpu, 0.9):7.3f}  p99 {pct(gpu, 0.99):7.3f}  "
          f"max {max(gpu):7.3f}  stalls(>1.3x) {stalls:3d}/{calls} | host queueing: median {pct(host, 0.5):6.3f}  max {max(host):7.3f} | "
          f"{pairs / (med * 1e-3):.3e} pairs/s at the median ({256 // B if 256 % B == 0 else 256 / B:g}-GPU shard; ideal share of B=256)", flush=True)
    return med


def first_calls(B, n, dev):
    """The first n calls on a fresh shape, no warm-up: event time, host queueing time and allocator activity of each."""
    L = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)
    x, y = bench.cfg4_batch(dev, B, seed=2)
    torch.cuda.synchronize()
    a = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    b = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    host, seg = [], []
    for k in range(n):
        s0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        a[k].record()
        h0 = time.perf_counter()
        L(x, y).sum()
        host.append((time.perf_counter() - h0) * 1e3)
        b[k].record()
        seg.append(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - s0)
    torch.cuda.synchronize()
    gpu = [s.elapsed_time(e) for s, e in zip(a, b)]
    print(f"B={B}: first {n} calls (event ms / host ms / device allocations): " +
          "  ".join(f"{g:.2f}/{h:.2f}/{m}" for g, h, m in zip(gpu, host, seg)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0, help="print the first N calls of every size one by one instead of the distribution")
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--sizes", default="256,128,64,32")
    ap.add_argument("--reverse", action="store_true")
    ap.add_argument("--no-fusion", action="store_true", help="four glhip_sinkhorn_step launches per iteration instead of one glhip_sinkhorn_iter4")
    args = ap.parse_args()
    if args.no_fusion:
        sinkhorn_samples.set_iteration_fusion(False)
    dev = torch.device("cuda:0")
    sizes = [int(s) for s in args.sizes.split(",")]
    if args.reverse:
        sizes = sizes[::-1]
    if args.first:
        for B in sizes:
            first_calls(B, args.first, dev)
        return
    meds = {B: probe(B, args.calls, dev) for B in sizes}
    if 256 in meds:
        for B in sizes:
            print(f"  shard efficiency B={B}: {meds[256] * B / 256 / meds[B]:.3f}")


if __name__ == "__main__":
    main()
