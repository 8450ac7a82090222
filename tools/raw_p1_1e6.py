"""GPU box: the raw C-ABI calls of the distance reductions at N = M = 1e6 (what INTEGRATION.md's stub binds), with the library's own
voxel sort (default) and without (GLHIP_FLAG_NO_SORT): ms per call, and the largest difference between the two answers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
g = torch.Generator().manual_seed(0)
x, y = torch.rand(1, n, 3, generator=g).to(dev), torch.rand(1, n, 3, generator=g).to(dev)
h = (0.3 * torch.randn(1, n, generator=g) - torch.log(torch.tensor(float(n)))).to(dev)
v = torch.full((1, n), 1.0 / n, device=dev)
def ms(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out
for name, call in (("glhip_softmin_fwd p=1", lambda fl: hip.softmin_fwd_raw(x, y, h, 0.05, 1, None, fl)),
                   ("glhip_sinkhorn_step p=1", lambda fl: hip.sinkhorn_step_raw(x, y, h, h * 0.01, h * 0.02, 0.05, 0.8, 1, None, fl)),
                   ("glhip_kernel_conv_fwd laplacian", lambda fl: hip.kernel_conv_fwd_raw(hip.LAPLACIAN, x, y, v, 0.05, None, fl)),
                   ("glhip_kernel_conv_fwd energy", lambda fl: hip.kernel_conv_fwd_raw(hip.ENERGY, x, y, v, 0.05, None, fl))):
    t1, a = ms(lambda: call(0))
    t2, b = ms(lambda: call(hip.FLAG_NO_SORT))
    print(f"{name:34s} self-sorting {t1:8.2f} ms   NO_SORT {t2:8.2f} ms   max |diff| / max |value| {((a - b).abs().max() / b.abs().max()).item():.2e}", flush=True)
