import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
def run(name, loss, n, backward, reps=2):
    g = torch.Generator().manual_seed(1)
    x = torch.rand(n, 3, generator=g).to(dev).requires_grad_(backward)
    y = torch.rand(n, 3, generator=g).to(dev)
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L = loss(x, y)
        if backward:
            torch.autograd.grad(L, [x])
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(name, ["%.4f" % t for t in ts], float(L))
n = 1_000_000
run("gauss_ms nograd", SamplesLoss("gaussian", blur=0.05, backend="multiscale"), n, False)
run("gauss_ms grad", SamplesLoss("gaussian", blur=0.05, backend="multiscale"), n, True)
run("gauss_online nograd", SamplesLoss("gaussian", blur=0.05, backend="online"), n, False)
run("gauss_ms nograd again", SamplesLoss("gaussian", blur=0.05, backend="multiscale"), n, False)
run("gauss_ms blur.1 trunc3 grad", SamplesLoss("gaussian", blur=0.1, truncate=3, backend="multiscale"), n, True)
run("gauss_online blur.1 grad", SamplesLoss("gaussian", blur=0.1, backend="online"), n, True)
