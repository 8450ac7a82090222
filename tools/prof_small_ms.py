"""Where a small multiscale loss spends its time: host time per library call (ctypes wrapper), cProfile, wall clock.
usage: python tools/prof_small_ms.py [N] [backend]     (under rocprofv3 --kernel-trace --stats for the device side)"""
import sys, os, time, cProfile, pstats, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss, hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000
backend = sys.argv[2] if len(sys.argv) > 2 else "multiscale"
x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend=backend)
for _ in range(5): loss(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): L = loss(x, y)
torch.cuda.synchronize()
print("ms/loss", (time.perf_counter() - t0) / 20 * 1e3)
if os.environ.get("HOST", "1") == "1":
    lib = hip.load_library()
    acc = collections.defaultdict(lambda: [0, 0.0])

    class Timed:
        def __init__(self, name, fn): self.name, self.fn = name, fn
        def __call__(self, *a):
            t = time.perf_counter(); r = self.fn(*a); e = acc[self.name]; e[0] += 1; e[1] += time.perf_counter() - t
            return r
    for name in hip.SIGNATURES:
        setattr(lib, name, Timed(name, getattr(lib, name)))
    for _ in range(20): L = loss(x, y)
    torch.cuda.synchronize()
    for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:36s} calls/loss {c/20:5.1f}   host us/call {t/c*1e6:8.1f}")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): L = loss(x, y)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
