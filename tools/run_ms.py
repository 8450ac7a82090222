"""GPU box: three (REPS) two-scale losses on uniform clouds, for kernel traces and timings.  usage: [REPS=30] run_ms.py N D"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
n, D = int(float(sys.argv[1])), int(sys.argv[2])
g = torch.Generator().manual_seed(3)
x, y = torch.rand(n, D, generator=g).to(dev), torch.rand(n, D, generator=g).to(dev)
loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend=os.environ.get("BACKEND", "multiscale"), verbose=len(sys.argv) > 3)      # BACKEND=online: the single-scale loss
reps, ts = int(os.environ.get("REPS", "3")), []       # REPS=30: the median over many calls (small clouds: +-0.2 ms from call to call)
for r in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L = loss(x, y)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    if reps <= 3:
        print(f"rep {r}: {ts[-1]:.2f} ms  loss {L.item():.6e}", flush=True)
if reps > 3:
    print(f"N = {n}: median {sorted(ts)[reps // 2]:.3f} ms, min {min(ts):.3f} ms over {reps} calls  loss {L.item():.6e}", flush=True)
