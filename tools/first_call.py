"""Where does the first call go?  Times, in a fresh process, (1) loading the HIP library and launching its first kernel,
(2) the first and second call of each phase of SamplesLoss("sinkhorn", backend="multiscale") at N = 1e6, fenced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def tick():
    torch.cuda.synchronize(); return time.perf_counter()

dev = torch.device("cuda:0")
t0 = time.perf_counter(); torch.zeros(1, device=dev); t1 = tick()
print(f"torch CUDA context: {(t1 - t0) * 1e3:.1f} ms")
from geomloss_amd import SamplesLoss, hip, sinkhorn_samples as ss
t0 = tick(); hip.load_library(); t1 = tick()
x = torch.rand(1, 256, 3, device=dev); h = torch.zeros(1, 256, device=dev)
hip.softmin_fwd_raw(x, x, h, 0.01, 2); t2 = tick()
hip.softmin_fwd_raw(x, x, h, 0.01, 2); t3 = tick()
print(f"dlopen libgeomloss_hip.so: {(t1 - t0) * 1e3:.1f} ms; first launch (code object load): {(t2 - t1) * 1e3:.1f} ms; second: {(t3 - t2) * 1e3:.2f} ms")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g = torch.Generator().manual_seed(1)
xs, ys = torch.rand(N, 3, generator=g).to(dev), torch.rand(N, 3, generator=g).to(dev)
import collections
acc = [collections.OrderedDict(), collections.OrderedDict()]
cur = [0]
def timed(name, fn):
    def wrapper(*a, **k):
        t = tick(); out = fn(*a, **k); acc[cur[0]][name] = acc[cur[0]].get(name, 0.0) + tick() - t
        return out
    return wrapper
ss.clusterize = timed("clusterize", ss.clusterize)
ss.kernel_truncation = timed("kernel_truncation", ss.kernel_truncation)
ss.extrapolate_samples = timed("extrapolate", ss.extrapolate_samples)
o_soft, o_step = hip.softmin, hip.sinkhorn_step
def softmin(eps, x_, y_, h, **kw):
    return timed("fine block-sparse" if kw.get("ranges") is not None else f"dense N={x_.shape[-2]} M={y_.shape[-2]}", o_soft)(eps, x_, y_, h, **kw)
def step(eps, x_, y_, logw, pot, prev, damping, **kw):
    return timed("fine block-sparse" if kw.get("ranges") is not None else f"dense N={x_.shape[-2]} M={y_.shape[-2]}", o_step)(eps, x_, y_, logw, pot, prev, damping, **kw)
hip.softmin, hip.sinkhorn_step = softmin, step
L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale")
tot = []
for k in range(2):
    cur[0] = k
    t = tick(); L(xs, ys); tot.append(tick() - t)
print(f"multiscale N={N}: first call {tot[0] * 1e3:.1f} ms, second {tot[1] * 1e3:.1f} ms")
for name in acc[0]:
    print(f"  {acc[0][name] * 1e3:9.2f} ms first | {acc[1].get(name, 0) * 1e3:9.2f} ms second   {name}")
print(f"  {(tot[0] - sum(acc[0].values())) * 1e3:9.2f} ms first | {(tot[1] - sum(acc[1].values())) * 1e3:9.2f} ms second   everything else")
