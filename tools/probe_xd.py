"""GPU box: the 4 <= D <= 16 matrix-core kernels at N = M = 1e6 (glhip_softmin_xd.h): soft-min forward and gaussian product,
milliseconds per launch (HIP events, median of 5).  GLHIP_XD_PRE=0 selects on-the-fly column packing (A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip

dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
dims = [int(d) for d in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 5, 8, 12, 16]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


print(f"GLHIP_XD_PRE={os.environ.get('GLHIP_XD_PRE', '(default 1)')}  N=M={n}")
for D in dims:
    g = torch.Generator().manual_seed(D)
    x = torch.rand(1, n, D, generator=g).to(dev)
    y = torch.rand(1, n, D, generator=g).to(dev)
    h = (torch.randn(1, n, generator=g) * 2).to(dev)
    v = (torch.rand(1, n, generator=g) / n).to(dev)
    eps = 0.05 ** 2 * D / 3
    t1 = timed(lambda: hip.softmin_fwd_raw(x, y, h, eps, 2))
    t2 = timed(lambda: hip.kernel_conv_fwd_raw(hip.GAUSSIAN, x, y, v, 0.05 * (D / 3) ** 0.5))
    cyc = lambda ms: ms * 1e-3 * 2.4e9 * 1024 / (float(n) * n / 1024)
    line = f"D={D:2d}: softmin fwd {t1:8.2f} ms ({cyc(t1):5.0f} cyc / 1024 pairs / SIMD)   gaussian product {t2:8.2f} ms ({cyc(t2):5.0f})"
    if os.environ.get("GRAD", "0") == "1":      # the gradient kernels (glhip_wsum_t32.h)
        out = hip.softmin_fwd_raw(x, y, h, eps, 2)
        g = torch.randn(1, n, device=dev)
        t3 = timed(lambda: hip.softmin_bwd_x_raw(x, y, h, out, g, eps, 2), reps=3)
        line += f"   softmin gradient {t3:8.2f} ms"
    print(line, flush=True)
