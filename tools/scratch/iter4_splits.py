"""GPU box: time of one glhip_sinkhorn_iter4 launch (4 soft-mins, D = 3, f16 x 2) against the clouds' size; run once per value of
GLHIP_ITER4_SPLITS (latched per process) to fit the split rule of small launches.  usage: iter4_splits.py [sizes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geomloss_amd import hip
dev = torch.device("cuda:0")
sizes = [int(float(s)) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1000,2000,5000,10000,20000,30000".split(","))]
g = torch.Generator().manual_seed(0)
row = []
for n in sizes:
    x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
    al = torch.full((n,), -float(torch.tensor(float(n)).log()), device=dev)
    plan = hip.Iter4Plan(x, y, al, al, True)
    plan.extra_flags = hip.FLAG_F16X2
    pots = plan.run(0.01, 1.0, None)
    for _ in range(5):
        pots = plan.run(0.01, 1.0, pots)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    a.record()
    for _ in range(reps):
        pots = plan.run(0.01, 1.0, pots)
    b.record(); torch.cuda.synchronize()
    row.append(f"{n}: {a.elapsed_time(b) / reps * 1e3:7.1f} us")
print(f"splits={os.environ.get('GLHIP_ITER4_SPLITS', 'rule'):>4s}  " + "   ".join(row), flush=True)
