"""Wall-clock of small / medium problems (launch-bound regime): median per-call time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")


def median_call(fn, reps):
    """Median of per-call wall-clock times: one call in a few dozen takes 1-35 ms more (a replayed graph re-uploading, the allocator
    trimming), which a mean over 5-10 calls turns into the headline of a 0.4-ms row."""
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


for N, D in ((2000, 2), (10000, 3), (30000, 3)):
    g = torch.Generator().manual_seed(0)
    x, y = torch.rand(N, D, generator=g).to(dev), torch.rand(N, D, generator=g).to(dev)
    for backend in ("online", "tensorized", "multiscale"):
        if backend == "tensorized" and N > 10000: continue
        L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend=backend)
        for _ in range(2): L(x, y)
        t = median_call(lambda: L(x, y), 7)
        v = L(x, y)
        print(f"N={N:6d} D={D} {backend:10s} {t*1e3:8.3f} ms/loss  loss={v.item():.6e}", flush=True)
    # hipGraph mode (needs a fixed diameter)
    from geomloss_amd import sinkhorn_samples as ss
    for mode in (False, True):
        ss.set_graph_mode(mode)
        L = SamplesLoss("sinkhorn", p=2, blur=0.05, diameter=1.8, backend="online")
        for _ in range(3): L(x, y)
        t = median_call(lambda: L(x, y), 11)
        v = L(x, y)
        print(f"N={N:6d} D={D} online diameter=1.8 graph={mode!s:5s} {t*1e3:8.3f} ms/loss  loss={v.item():.6e}", flush=True)
    ss.set_graph_mode(False)
