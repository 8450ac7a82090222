import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss, hip
dev = torch.device("cuda:0")
def run(N, D, backend, reps=300, **kw):
    g = torch.Generator().manual_seed(1)
    x, y = torch.rand(N, D, generator=g).to(dev), torch.rand(N, D, generator=g).to(dev)
    L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend=backend, **kw)
    for _ in range(20): L(x, y)
    hip.settle_host()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): L(x, y)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    pr = cProfile.Profile(); pr.enable()
    for _ in range(reps): L(x, y)
    torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
    print(f"==== N={N} D={D} {backend} {kw}: {ms:.3f} ms/loss"); print("\n".join(l[:150] for l in s.getvalue().split("\n")[4:30]))
run(10000, 3, "multiscale")
run(2000, 2, "online")
