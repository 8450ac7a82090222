import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from geomloss_amd import hip, SamplesLoss
dev = torch.device("cuda:0")
x, y, h, eps = bench.make_problem(1_000_000, dev, seed=1000)
print("headline f16x2: %.2f ms   bf16x3: %.2f ms" % (bench.event_ms(lambda: hip.softmin_fwd_raw(x, y, h, eps, 2, flags=256), 5),
                                                  bench.event_ms(lambda: hip.softmin_fwd_raw(x, y, h, eps, 2), 3)), flush=True)
g = torch.Generator().manual_seed(1)
for n, be, reps in ((1_000_000, "multiscale", 3), (100_000, "multiscale", 5), (100_000, "online", 3), (10_000, "multiscale", 10)):
    xs, ys = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
    L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend=be)
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter(); v = L(xs, ys); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{be} {n}: {min(ts[1:])*1e3:.3f} ms  loss {v.item():.9e}", flush=True)
for B in (256, 32):
    xb, yb = bench.cfg4_batch(dev, B, seed=2)
    L = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)
    for _ in range(3): L(xb, yb)
    print(f"cfg4 B={B}: {bench.event_ms(lambda: L(xb, yb).sum(), 10):.3f} ms", flush=True)
