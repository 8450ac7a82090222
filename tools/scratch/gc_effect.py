"""GPU box: per-call time of a small online loss with the cyclic collector on / off / frozen (is the ~150 us that cProfile pins on
the first torch call after the annealing launches a generation-0/1 collection?).  usage: gc_effect.py [N]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = torch.Generator().manual_seed(3)
x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online")


def med(reps=200):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss(x, y)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[reps // 2] * 1e3, ts[int(reps * 0.9)] * 1e3, sum(ts) / reps * 1e3


for _ in range(30):
    loss(x, y)
print("gc on      median %.3f  p90 %.3f  mean %.3f ms" % med(), gc.get_count(), gc.get_threshold())
gc.collect(); gc.freeze()
print("gc frozen  median %.3f  p90 %.3f  mean %.3f ms" % med())
gc.disable()
print("gc off     median %.3f  p90 %.3f  mean %.3f ms" % med())
gc.enable()
# host time only: no synchronisation inside the loop
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    loss(x, y)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("200 calls back to back: host %.3f ms/call, with the final sync %.3f ms/call" % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
