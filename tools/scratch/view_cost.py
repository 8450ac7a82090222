"""GPU box: why do the four .view calls at the end of Iter4Plan.run(last=True) take 40 us each under cProfile?  Times the segments
of that function (allocation, library call, views) with perf_counter, in a loop like the loss's: anneal + last run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geomloss_amd import hip
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = torch.Generator().manual_seed(3)
x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
al = torch.full((n,), -6.9, device=dev)
eps_list = [3.0 * 0.25 ** k for k in range(10)]
damp = [1.0] * 10
acc = {"alloc": 0.0, "call": 0.0, "views": 0.0, "anneal": 0.0, "plan": 0.0, "sync": 0.0}
reps = 300
for r in range(reps + 20):
    if r == 20:
        acc = {k: 0.0 for k in acc}
    t0 = time.perf_counter()
    plan = hip.Iter4Plan(x, y, al, al, True)
    t1 = time.perf_counter()
    new, old = plan.anneal(eps_list, damp, 0.01)
    t2 = time.perf_counter()
    B, N, M, D = plan.dims
    outs = [torch.empty((B, k), dtype=torch.float32, device=dev) for k in (N, M, N, M)]
    t3 = time.perf_counter()
    ptr = tuple(t.data_ptr() for t in new)
    rc = plan.lib.glhip_sinkhorn_iter4(*plan.fixed, *ptr, *(t.data_ptr() for t in outs), B, N, M, D, 0.01, 1.0, 2, plan.dtype, 2,
                                       plan.ws.data_ptr(), plan.nbytes, 0, torch.cuda.current_stream(dev).cuda_stream)
    t4 = time.perf_counter()
    res = tuple(t.view(sh) for t, sh in zip(outs, plan.shapes))
    t5 = time.perf_counter()
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    for k, v in zip(("plan", "anneal", "alloc", "call", "views", "sync"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
        acc[k] += v
print("  ".join(f"{k} {v / reps * 1e6:.1f} us" for k, v in acc.items()))
