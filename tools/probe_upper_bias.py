"""Per-term rounding bias of the kernel-norm self-term <a, K_xx a> at N = 1e6: full product vs upper triangle, in the forward
family (flags 0) and the gradient family (GRAD_FAMILY), against float64.  What matters for a loss between two samples of one law
(1e-6 of its terms) is that the three terms carry the SAME relative bias."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geomloss_amd.kernel_samples as ks
from geomloss_amd import hip
from oracle import oracle_torch64 as o64
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
g = torch.Generator().manual_seed(13)
x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
a = torch.full((n,), 1.0 / n, device=dev)
if os.environ.get("SORT", "0") == "1":      # both clouds along the boustrophedon voxel path: the row blocks become spatially compact
    x, y = hip.compact_order(x)[1].contiguous(), hip.compact_order(y)[1].contiguous()
    print("clouds in compact order")
for name in os.environ.get("KERNELS", "gaussian laplacian energy").split():
    ref = {}
    for tag, (p, q) in (("xx", (x, x)), ("yy", (y, y)), ("xy", (x, y))):
        ref[tag] = float((a.double().cpu().numpy() * o64.kconv(name, p, q, a, 0.05, device=dev)).sum())
    for fam, flags in (("fwd ", 0), ("grad", hip.FLAG_GRAD_FAMILY)):
        out = {}
        for tag, (p, q) in (("xx", (x, x)), ("yy", (y, y)), ("xy", (x, y))):
            out[tag] = (a * hip.kernel_conv(name, p, q, a, 0.05, flags=flags)).double().sum().item()
        if flags == 0:      # the upper-triangle path runs in the forward family only
            for tag, p in (("xx", x), ("yy", y)):
                out[tag + "_upper"] = ks._self_term_value(name, p, a, 0.05).double().item()
        xg = x.clone().requires_grad_(True)
        out["xy_fused"] = (a * hip.kernel_conv(name, xg, y, a, 0.05, flags=flags)).double().sum().item()
        out["xx_fused"] = (a * hip.kernel_conv(name, xg, x, a, 0.05, flags=flags)).double().sum().item()
        line = "  ".join(f"{k} {(v - ref[k[:2]]) / abs(ref[k[:2]]):+.2e}" for k, v in out.items())
        print(f"{name:10s} {fam} relative error of the term:  {line}", flush=True)
    print(f"{name:10s} terms (fp64): xx {ref['xx']:.9e}  yy {ref['yy']:.9e}  xy {ref['xy']:.9e}  loss {0.5 * (ref['xx'] + ref['yy']) - ref['xy']:.6e}", flush=True)
