import sys, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import torch
from geomloss_amd import SamplesLoss
dev = torch.device('cuda:0')
torch.manual_seed(0)
N = 1000
x = torch.rand(N, 3, device=dev, requires_grad=True); y = torch.rand(N, 3, device=dev)
L = SamplesLoss("sinkhorn", p=2, blur=0.05, diameter=1.0, backend="online")
def step():
    l = L(x, y); g, = torch.autograd.grad(l, [x]); return l
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print("ms per loss+backward:", (time.perf_counter() - t0) / 200 * 1e3)
with torch.no_grad():
    xd = x.detach()
    for _ in range(20): L(xd, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): L(xd, y)
    torch.cuda.synchronize(); print("ms per loss (no grad):", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
