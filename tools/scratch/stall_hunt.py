"""Where does the intermittent +20 ms of the 10-call B = 32 loop (tools/probe_batch.py) live: GPU timeline or host wake-up?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
L = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)
mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ncall = int(sys.argv[3]) if len(sys.argv) > 3 else 10
import gc
gc_log = []
def _cb(phase, info, _t=[0.0]):
    if phase == "start": _t[0] = time.perf_counter()
    else: gc_log.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3))
gc.callbacks.append(_cb)
if os.environ.get("STALL_GC") == "off": gc.disable()
if os.environ.get("STALL_GC") == "freeze": gc.collect(); gc.freeze()
x, y = bench.cfg4_batch(dev, B, seed=2)
for _ in range(3): L(x, y)
rows = []
for rep in range(30):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(ncall): L(x, y).sum()
    e1.record()
    tq = time.perf_counter()
    if mode == "sync":
        torch.cuda.synchronize()
    elif mode == "spin":
        while not e1.query(): pass
    elif mode == "evsync":
        e1.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    rows.append(((t1 - t0) * 1e3 / ncall, e0.elapsed_time(e1) / ncall, (tq - t0) * 1e3 / ncall))
print(f"mode={mode} B={B} calls={ncall}: per-call wall / events / host-queue ms over 30 blocks:")
print("  " + "  ".join(f"{w:.2f}/{e:.2f}/{q:.2f}" for w, e, q in rows))
print("  gc: gen2 collections (ms):", [round(t, 1) for g, t in gc_log if g == 2], " gen1:", len([1 for g, t in gc_log if g == 1]), " gen0:", len([1 for g, t in gc_log if g == 0]),
      " longest gen0/1: %.2f ms" % max([t for g, t in gc_log if g < 2] + [0.0]))
