import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geomloss_amd import SamplesLoss
from geomloss_amd.sinkhorn_divergence import log_weights, scaling_parameters, sinkhorn_cost, sinkhorn_loop
from geomloss_amd import sinkhorn_samples as ss
dev = torch.device("cuda:0")
B = 32
x, y = bench.cfg4_batch(dev, B, seed=2)
a = torch.full((B, 4096), 1 / 4096, device=dev); b = a.clone()
def timeit(name, fn, n=24):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:34s}", " ".join(f"{t:.1f}" for t in ts), flush=True)
timeit("log_weights x2", lambda: (log_weights(a), log_weights(b)))
al, bl = log_weights(a), log_weights(b)
diam, eps, eps_list, rho = scaling_parameters(x, y, 2, 0.05, None, 1.8, 0.5)
sm = ss._HipSoftmin(2, False)
def loop():
    s = ss._HipSoftmin(2, False)
    return sinkhorn_loop(s, al, bl, (x, x), (y, y), (x, y), (y, x), eps_list, rho, debias=True)
timeit("sinkhorn_loop", loop)
pots = loop()
timeit("sinkhorn_cost", lambda: sinkhorn_cost(eps, rho, a, b, *pots, batch=True, debias=True))
L = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)
timeit("SamplesLoss call", lambda: L(x, y))
timeit("SamplesLoss call (a,x,b,y)", lambda: L(a, x, b, y))
