"""GPU box: the segments of Iter4Plan.run(last=True) timed IN the loss (cProfile pins ~150 us on its first torch call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geomloss_amd import SamplesLoss, hip
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = torch.Generator().manual_seed(3)
x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online")
acc = {}
orig_run, orig_anneal, orig_apply = hip.Iter4Plan.run, hip.Iter4Plan.anneal, hip._Last4.apply


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


hip.Iter4Plan.run = timed("run", orig_run)
if os.environ.get("SYNC_AFTER_ANNEAL"):
    def synced(*a, **k):
        r = orig_anneal(*a, **k)
        torch.cuda.synchronize()
        return r
    hip.Iter4Plan.anneal = timed("anneal+sync", synced)
else:
    hip.Iter4Plan.anneal = timed("anneal", orig_anneal)
hip.sinkhorn_last4 = timed("last4", hip.sinkhorn_last4)
lib = hip.load_library()
lib.glhip_sinkhorn_iter4 = timed("C iter4", lib.glhip_sinkhorn_iter4)
lib.glhip_sinkhorn_anneal = timed("C anneal", lib.glhip_sinkhorn_anneal)
hip._check = timed("_check", hip._check)
torch.cuda.current_stream = timed("current_stream", torch.cuda.current_stream)
real_empty = torch.empty
torch.empty = timed("empty", real_empty)
for r in range(320):
    if r == 20:
        acc.clear(); t_all = time.perf_counter()
    loss(x, y)
tot = time.perf_counter() - t_all
print("per loss: total %.1f us | " % (tot / 300 * 1e6) + "  ".join(f"{k} {v / 300 * 1e6:.1f} us" for k, v in acc.items()))
