"""Kernel norms, forward + backward, with / without the one-pass product-and-gradient kernels; and the same-law accuracy
(two samples of one law: the three terms cancel to ~1e-5 of their size) against the fp64 oracle."""
import sys
import time

import torch

sys.path.insert(0, ".")
from geomloss_amd import SamplesLoss, hip  # noqa: E402
from oracle import oracle_torch64  # noqa: E402

dev = torch.device("cuda:0")


def clouds(N, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand(N, 3, generator=g).to(dev)
    y = torch.rand(N, 3, generator=g).to(dev)
    return x, y


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for N in (100_000, 1_000_000):
    x, y = clouds(N)
    for name in ("gaussian", "laplacian", "energy"):
        L = SamplesLoss(name, blur=0.05, backend="online")
        row = []
        for fused in (True, False):
            hip.set_kernel_grad_fusion(fused)

            def fb():
                xg = x.clone().requires_grad_(True)
                (g,) = torch.autograd.grad(L(xg, y), [xg])
                return g

            row.append(timed(fb))
        hip.set_kernel_grad_fusion(True)
        fwd = timed(lambda: L(x, y))
        print(f"N={N} {name:10s} fwd {fwd*1e3:8.1f} ms   fwd+bwd one-pass {row[0]*1e3:8.1f} ms   two-pass {row[1]*1e3:8.1f} ms", flush=True)

# same-law accuracy at 1e5 (oracle: chunked fp64 on the GPU)
N = 100_000
x, y = clouds(N, seed=3)
for name in ("gaussian", "laplacian", "energy"):
    L = SamplesLoss(name, blur=0.05, backend="online")
    ref = oracle_torch64.kernel_loss(name, x, y, blur=0.05) if hasattr(oracle_torch64, "kernel_loss") else None
    ref_v = float(ref[0] if isinstance(ref, tuple) else ref)
    out = {}
    for fused in (True, False):
        hip.set_kernel_grad_fusion(fused)
        xg = x.clone().requires_grad_(True)
        v = L(xg, y)
        (g,) = torch.autograd.grad(v, [xg])
        out[fused] = (v.item(), g)
    hip.set_kernel_grad_fusion(True)
    v0 = L(x, y).item()
    gd = (out[True][1] - out[False][1]).abs().max().item() / out[False][1].abs().max().item()
    print(f"same-law {name:10s} ref {ref_v:.6e}  no-grad {v0:.6e} ({abs(v0-ref_v)/abs(ref_v):.1e})  one-pass {out[True][0]:.6e} "
          f"({abs(out[True][0]-ref_v)/abs(ref_v):.1e})  two-pass {out[False][0]:.6e} ({abs(out[False][0]-ref_v)/abs(ref_v):.1e})  grad diff {gd:.1e}", flush=True)
