"""Per-phase wall-clock of SamplesLoss("sinkhorn", backend="multiscale"): clustering, coarse-level soft-mins,
kernel truncation (keep-mask -> ranges), extrapolations, fine-level block-sparse soft-mins.  Phases are fenced with
torch.cuda.synchronize(), so their sum exceeds the un-instrumented loss time (which overlaps host and device work)."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss, sinkhorn_samples as ss, hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x, y = torch.rand(N, 3, generator=g).to(dev), torch.rand(N, 3, generator=g).to(dev)
L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale")
for _ in range(2):
    L(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter(); L(x, y); torch.cuda.synchronize(); plain = time.perf_counter() - t0

acc = collections.OrderedDict()
def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return out
    return wrapper

ss.clusterize = timed("clusterize (voxel labels, sort, ranges, centroids)", ss.clusterize)
ss.kernel_truncation = timed("kernel_truncation (coarse keep-mask -> fine ranges)", ss.kernel_truncation)
ss.extrapolate_samples = timed("extrapolate (fine x coarse dense soft-min)", ss.extrapolate_samples)
orig_softmin, orig_step = hip.softmin, hip.sinkhorn_step
def softmin(eps, x_, y_, h, p=2, ranges=None, flags=0):
    name = "fine block-sparse soft-min" if ranges is not None else ("coarse / dense soft-min (N=%d)" % x_.shape[-2])
    return timed(name, orig_softmin)(eps, x_, y_, h, p=p, ranges=ranges, flags=flags)
def step(eps, x_, y_, logw, pot, prev, damping, p=2, ranges=None, flags=0):
    name = "fine block-sparse soft-min" if ranges is not None else ("coarse / dense soft-min (N=%d)" % x_.shape[-2])
    return timed(name, orig_step)(eps, x_, y_, logw, pot, prev, damping, p=p, ranges=ranges, flags=flags)
hip.softmin, hip.sinkhorn_step = softmin, step
t0 = time.perf_counter(); L(x, y); torch.cuda.synchronize(); total = time.perf_counter() - t0
print(f"N=M={N}: un-instrumented loss {plain*1e3:.1f} ms; instrumented {total*1e3:.1f} ms")
for k, v in acc.items():
    print(f"  {v*1e3:9.2f} ms  {k}")
print(f"  {(total - sum(acc.values()))*1e3:9.2f} ms  everything else (eps schedule, diameter, loop glue, sinkhorn_cost)")
