"""Generates tests/golden/images_*.npz by running the REFERENCE image / volume path (jeanfeydy/geomloss 0.3.1,
/root/reference: `_legacy/sinkhorn_images.py`, `_legacy/wasserstein_barycenter_images.py`, `_legacy/utils.py`).

    python tests/golden/make_golden_images.py

That path needs pykeops for ONE primitive — `LazyTensor(...).logsumexp(dim=2)` inside `softmin_grid`
(`_legacy/utils.py:254-270`) — and pykeops is not installed here.  The script therefore hands the reference a dense
stand-in for `pykeops.torch.LazyTensor` (broadcasting torch tensors: `-`, `abs`, `**`, `logsumexp`), which evaluates
the very same formula; everything else — pyramid, temperatures and jumps, the loop, interpolation, the loss and the
barycenter iteration — is the reference's own code.  Only this script reads /root/reference.
"""

import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/src")
import geomloss  # noqa: E402  (the reference)
from geomloss._legacy import utils as ref_utils  # noqa: E402

assert geomloss.__version__ == "0.3.1"
OUT = os.path.dirname(os.path.abspath(__file__))


class DenseLazy:
    """Just enough of pykeops.torch.LazyTensor for softmin_grid: symbolic (.., N, N, 1) arrays held densely."""

    def __init__(self, t):
        self.t = t

    def __sub__(self, other):
        return DenseLazy(self.t - other.t)

    def abs(self):
        return DenseLazy(self.t.abs())

    def __pow__(self, k):
        return DenseLazy(self.t**k)

    def logsumexp(self, dim):
        return self.t.logsumexp(dim)


ref_utils.keops_available = True
ref_utils.LazyTensor = DenseLazy

from geomloss._legacy.sinkhorn_images import sinkhorn_divergence  # noqa: E402
from geomloss._legacy.wasserstein_barycenter_images import ImagesBarycenter  # noqa: E402


def densities(seed, shape, dtype=torch.float64, holes=True):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(shape, generator=g, dtype=dtype) ** 3
    if holes:
        a[a < 0.05] = 0.0                      # empty pixels: exercises log_dens' -10000
    return a / a.flatten(2).sum(-1).view(shape[0], shape[1], *([1] * (len(shape) - 2)))


SINKHORN = {
    # name: (kwargs, seed, shape)
    "images_p2_16": (dict(p=2), 1, (2, 1, 16, 16)),
    "images_p2_32_blur": (dict(p=2, blur=0.04, scaling=0.7), 2, (1, 2, 32, 32)),
    "images_p1_16": (dict(p=1, blur=0.1, scaling=0.6), 3, (2, 1, 16, 16)),
    "images_p2_reach": (dict(p=2, blur=0.1, reach=0.4), 4, (1, 1, 16, 16)),
    "images_p2_nodebias": (dict(p=2, blur=0.1, debias=False), 5, (1, 1, 16, 16)),
    "volumes_p2_8": (dict(p=2), 6, (1, 1, 8, 8, 8)),
}


def main():
    for name, (kw, seed, shape) in SINKHORN.items():
        rec = {}
        for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            a, b = densities(seed, shape).to(dtype), densities(seed + 100, shape).to(dtype)
            ag = a.clone().requires_grad_(True)
            L = sinkhorn_divergence(ag, b, **kw)
            (ga,) = torch.autograd.grad(L.sum(), [ag])
            F, G = sinkhorn_divergence(a, b, potentials=True, **kw)
            rec.update({f"loss_{tag}": L.detach().numpy(), f"ga_{tag}": ga.numpy(), f"F_{tag}": F.detach().numpy(),
                        f"G_{tag}": G.detach().numpy()})
            if dtype == torch.float64:
                rec["a"], rec["b"] = a.numpy(), b.numpy()
        rec["kwargs"] = np.array(repr(kw))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, rec["loss_f64"], rec["loss_f32"])

    # raw separable soft-min
    g = torch.Generator().manual_seed(9)
    rec = {}
    for p, eps, shape in ((2, 0.01, (2, 3, 16, 16)), (1, 0.05, (1, 2, 8, 8)), (2, 0.003, (1, 1, 8, 8, 8))):
        h = torch.randn(shape, generator=g, dtype=torch.float64) * 3
        h[..., 0] = -10000.0
        rec[f"h_p{p}_{len(shape)}"] = h.numpy()
        rec[f"out_p{p}_{len(shape)}"] = ref_utils.softmin_grid(eps, p, h).numpy()
        rec[f"eps_p{p}_{len(shape)}"] = np.array(eps)
    np.savez_compressed(os.path.join(OUT, "images_softmin_grid.npz"), **rec)

    # barycenter, with a gradient through the last iterations
    for name, (p, blur, scaling_N, K, N) in {"barycenter_p2_16": (2, 0, 4, 3, 16), "barycenter_p2_8_blur": (2, 0.2, 3, 2, 8)}.items():
        rec = {}
        for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            m = densities(21, (2, K, N, N)).to(dtype)
            w = torch.tensor([[0.2, 0.3, 0.5][:K], [0.6, 0.1, 0.3][:K]], dtype=dtype)
            w = (w / w.sum(1, keepdim=True)).requires_grad_(True)
            target = densities(33, (2, 1, N, N), holes=False).to(dtype)
            # the reference builds the pyramid under no_grad, so only the barycentric weights receive gradients
            bar = ImagesBarycenter(m, w, blur=blur, p=p, scaling_N=scaling_N, backward_iterations=2)
            (gm,) = torch.autograd.grad((bar * target).sum(), [w])
            rec.update({f"bar_{tag}": bar.detach().numpy(), f"gw_{tag}": gm.numpy()})
            if dtype == torch.float64:
                rec.update(measures=m.numpy(), weights=w.detach().numpy(), target=target.numpy())
        rec["cfg"] = np.array([p, blur, scaling_N])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, float(rec["bar_f64"].sum()), float(np.abs(rec["gw_f64"]).max()))


if __name__ == "__main__":
    main()
