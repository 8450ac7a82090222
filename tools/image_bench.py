"""Wall-clock of the grid path: sinkhorn_divergence on images / volumes and ImagesBarycenter (GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import ImagesBarycenter, sinkhorn_divergence, hip
dev = torch.device("cuda:0")
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
g = torch.Generator().manual_seed(0)
for shape in ((8, 1, 64, 64), (8, 1, 256, 256), (4, 1, 512, 512), (2, 1, 64, 64, 64), (1, 1, 128, 128, 128)):
    a = torch.rand(shape, generator=g).to(dev) ** 3; b = torch.rand(shape, generator=g).to(dev) ** 3
    dims = tuple(range(2, len(shape)))
    a, b = a / a.sum(dims, keepdim=True), b / b.sum(dims, keepdim=True)
    from geomloss_amd import sinkhorn_samples as ss
    t_eager = timeit(lambda: sinkhorn_divergence(a, b))
    ss.set_graph_mode(True)
    t_graph = timeit(lambda: sinkhorn_divergence(a, b))
    ss.set_graph_mode(False)
    print("sinkhorn_divergence %-22s %8.2f ms   (hipGraph replay: %.2f ms)" % (str(shape), t_eager, t_graph))
    N = shape[-1]; h = torch.randn(shape, generator=g).to(dev)
    t = timeit(lambda: hip.lse_lines(h, (1.0 / N) ** 2, 2), reps=20)
    print("   one line pass: %.3f ms = %.2e pairs/s" % (t, h.numel() * N / (t * 1e-3)))
m = torch.rand(4, 3, 128, 128, generator=g).to(dev) ** 3; m = m / m.sum((2, 3), keepdim=True)
w = torch.tensor([[0.2, 0.3, 0.5]] * 4, device=dev)
print("ImagesBarycenter (4,3,128,128), scaling_N=10: %8.2f ms" % timeit(lambda: ImagesBarycenter(m, w), reps=3))
