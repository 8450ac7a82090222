"""Bias of the three products of a kernel norm under the different arithmetic families, against the float64 HIP oracle:
<a, K_xx a>, <b, K_yy b>, <a, K_xy b> for the laplacian / energy kernels at N = M = n."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip
from oracle import oracle_hip64
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
w = torch.full((n,), 1.0 / n, device=dev)
for kind, blur in (("laplacian", 0.05), ("energy", 1.0)):
    ref = {}
    for name, (r, c) in dict(xx=(x, x), yy=(y, y), xy=(x, y)).items():
        ref[name] = float(oracle_hip64.kconv(kind, r, c, w.double(), blur, device=dev).sum() / n)
    loss_ref = 0.5 * ref["xx"] + 0.5 * ref["yy"] - ref["xy"]
    print(f"{kind}: float64 terms xx {ref['xx']:.9e} yy {ref['yy']:.9e} xy {ref['xy']:.9e} loss {loss_ref:.6e}")
    def report(label, fn):
        got = {name: float(fn(r, c).double().sum() / n) for name, (r, c) in dict(xx=(x, x), yy=(y, y), xy=(x, y)).items()}
        loss = 0.5 * got["xx"] + 0.5 * got["yy"] - got["xy"]
        print(f"  {label:46s}: rel err xx {got['xx']/ref['xx']-1:+.2e} yy {got['yy']/ref['yy']-1:+.2e} xy {got['xy']/ref['xy']-1:+.2e}  loss {loss:.6e} "
              f"(rel {loss/loss_ref-1:+.2e})", flush=True)
    report("matrix-core distances, sqrt", lambda r, c: hip.kernel_conv(kind, r, c, w, blur))
    report("matrix-core distances, FAMILY product", lambda r, c: hip.kernel_conv(kind, r, c, w, blur, flags=hip.FLAG_GRAD_FAMILY))
    def fused(r, c):
        rr = r.clone().requires_grad_(True)
        return hip.kernel_conv(kind, rr, c, w, blur, flags=hip.FLAG_GRAD_FAMILY).detach()
    report("matrix-core distances, product + gradient", fused)
    hip.set_distance_on_mfma(False)
    report("direct differences, sqrt", lambda r, c: hip.kernel_conv(kind, r, c, w, blur))
    report("direct differences, FAMILY product", lambda r, c: hip.kernel_conv(kind, r, c, w, blur, flags=hip.FLAG_GRAD_FAMILY))
    report("direct differences, product + gradient", fused)
    hip.set_distance_on_mfma(True)
