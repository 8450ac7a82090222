"""Generates tests/golden/multiscale_*.npz and kernel_multiscale_*.npz by running the REFERENCE's own two-scale drivers
(jeanfeydy/geomloss 0.3.1, /root/reference):

    `SamplesLoss("sinkhorn", backend="multiscale")`  ->  `_legacy/sinkhorn_samples.py:547-681` (`sinkhorn_multiscale`, with
        `clusterize :453-490`, `kernel_truncation :493-530`, `extrapolate_samples :533-544`, `softmin_multiscale :445-450`)
        driving `_legacy/sinkhorn_divergence.py:258-628` (`sinkhorn_loop`) and `:171-250` (`sinkhorn_cost`);
    `SamplesLoss("gaussian" | "laplacian" | "energy", backend="multiscale")`  ->  `_legacy/kernel_samples.py:177-271`
        (`kernel_multiscale`) and `:92-146` (`kernel_loss`).

    python tests/golden/make_golden_multiscale.py [case names]

Those drivers need pykeops for a handful of primitives, and pykeops is neither installed nor vendored here.  The script
hands the reference DENSE stand-ins for exactly those primitives — everything else (schedule, jump index, clustering
calls, the keep rule, the loop, the extrapolation, the loss, the un-permutation of the potentials, autograd) is the
reference's own code, imported from /root/reference.  Only this script reads /root/reference.

  primitive (pykeops)                      stand-in
  ---------------------------------------  -----------------------------------------------------------------------------
  generic_logsumexp(formula, aliases)      dense LSE_j of `B - P * cost(X, Y)` over an (N, M) matrix; `ranges=` (the KeOps
                                           6-tuple) is expanded to a point-level mask and the masked entries are -inf
  LazyTensor                               broadcasting torch tensors with `.ranges`, `@`, `.t()`, `sum`, `exp`, `sqrt`
  grid_cluster(x, size)                    voxel labels packed as sum_d floor(x_d / size) * 2^(10 (D-1-d)), compacted to
                                           0..C-1 in the order of the packed code (pykeops.torch.cluster, from memory —
                                           SURVEY.md Appendix B; the ORDER of clusters differs from oracle_np.grid_cluster
                                           on clouds with negative coordinates, the clusters themselves do not)
  cluster_ranges_centroids(x, lab, w)      bincount ranges, weighted centroids, summed weights
  sort_clusters((a, x), lab)               torch.sort(lab) and the permuted arrays
  from_matrix(ranges_i, ranges_j, keep)    the KeOps 6-tuple (ranges_i, slices_i, redranges_j, ranges_j, slices_j, redranges_i)
  swap_axes(r)                             (r[3], r[4], r[5], r[0], r[1], r[2])

`Norm2(X-Y)` (p = 1) and `LazyTensor.sqrt()` (laplacian, energy): the fixtures' primary values use the TENSORIZED backend's clamp `sqrt(max(d2, 1e-8))`
(`_legacy/utils.py:56-61`) — BASELINE.json's parity target is the tensorized backend, and that is what the HIP kernels
implement; the value with KeOps' un-clamped `Norm2` is stored beside it as `loss_f64_keops_norm2` (it differs by ~2e-4
relative at N = 600: the clamp turns the zero self-distances of the debiasing terms into 1e-4; 1e-5..1e-4 for the
laplacian and energy norms).
"""

import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/src")
import geomloss  # noqa: E402  (the reference)
from geomloss._legacy import kernel_samples as ref_ks  # noqa: E402
from geomloss._legacy import sinkhorn_samples as ref_ss  # noqa: E402
from geomloss._legacy import utils as ref_utils  # noqa: E402

assert geomloss.__version__ == "0.3.1"
OUT = os.path.dirname(os.path.abspath(__file__))
CLAMP_NORM2 = True          # see the module docstring


# ---- cluster helpers (pykeops.torch.cluster semantics) ---------------------------------------------------------------

def grid_cluster(x, size):
    with torch.no_grad():
        D = x.shape[1]
        if D > 3:
            raise NotImplementedError()
        weights = torch.tensor([2 ** (10 * (D - 1 - d)) for d in range(D)], dtype=torch.int64)
        lab = ((x / size).floor().to(torch.int64) * weights).sum(1)
        lab = lab - lab.min()
        u = torch.unique(lab)                       # sorted
        return torch.searchsorted(u, lab).to(torch.int32)


def cluster_ranges_centroids(x, lab, weights=None):
    lab = lab.long()
    C = int(lab.max()) + 1
    w = torch.ones(len(x), dtype=x.dtype) if weights is None else weights.view(-1)
    n_c = torch.bincount(lab, minlength=C)
    ends = torch.cumsum(n_c, 0)
    ranges = torch.stack([ends - n_c, ends], 1).int()
    w_c = torch.bincount(lab, weights=w, minlength=C)
    x_c = torch.stack([torch.bincount(lab, weights=w * x[:, d], minlength=C) for d in range(x.shape[1])], 1) / w_c[:, None]
    return ranges, x_c, w_c


PERMS = []          # the permutations sort_clusters applied (kernel_multiscale returns its potentials in SORTED order)


def sort_clusters(x, lab):
    lab_s, perm = torch.sort(lab.view(-1))
    PERMS.append(perm)
    if isinstance(x, tuple):
        return tuple(t[perm] for t in x), lab_s
    return x[perm], lab_s


def from_matrix(ranges_i, ranges_j, keep):
    I, J = torch.meshgrid(torch.arange(keep.shape[0]), torch.arange(keep.shape[1]), indexing="ij")
    redranges_j = ranges_j[J[keep]]                         # row-major over the kept (i, j)
    slices_i = keep.sum(1).cumsum(0).int()
    redranges_i = ranges_i[I.t()[keep.t()]]
    slices_j = keep.sum(0).cumsum(0).int()
    return (ranges_i, slices_i, redranges_j, ranges_j, slices_j, redranges_i)


def swap_axes(r):
    return (r[3], r[4], r[5], r[0], r[1], r[2])


_MASKS = {}


def expand_ranges(ranges, N, M):
    """KeOps block-sparse semantics: rows ranges_i[k] reduce over the union of redranges_j[slices_i[k-1]:slices_i[k]]."""
    key = (id(ranges[2]), N, M)
    if key not in _MASKS:
        ranges_i, slices_i, redranges_j = (t.long() for t in ranges[:3])
        mask = torch.zeros(N, M, dtype=torch.bool)
        start = 0
        for k in range(len(ranges_i)):
            cols = torch.zeros(M, dtype=torch.bool)
            for s in range(start, int(slices_i[k])):
                cols[redranges_j[s, 0]:redranges_j[s, 1]] = True
            mask[ranges_i[k, 0]:ranges_i[k, 1]] = cols
            start = int(slices_i[k])
        _MASKS.clear()
        _MASKS[key] = (ranges[2], mask)          # holds the tensor, so that its id stays unique
    return _MASKS[key][1]


# ---- reductions ------------------------------------------------------------------------------------------------------

class _KeopsSqrt(torch.autograd.Function):
    """sqrt with KeOps' convention for the derivative at 0 (Rsqrt(0) = 0)."""

    @staticmethod
    def forward(ctx, t):
        r = t.sqrt()
        ctx.save_for_backward(r)
        return r

    @staticmethod
    def backward(ctx, g):
        (r,) = ctx.saved_tensors
        return g * torch.where(r > 0, 0.5 / torch.where(r > 0, r, torch.ones_like(r)), torch.zeros_like(r))


def generic_logsumexp(formula, *aliases):
    assert aliases == ("A = Vi(1)", aliases[1], aliases[2], "B = Vj(1)", "P = Pm(1)"), aliases
    if formula == "( B - (P * (SqDist(X,Y) / IntCst(2)) ) )":
        cost = lambda x, y: ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1) / 2          # noqa: E731
    elif formula == "( B - (P * Norm2(X-Y) ) )":
        if CLAMP_NORM2:
            cost = lambda x, y: ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1).clamp_min(1e-8).sqrt()      # noqa: E731
        else:
            cost = lambda x, y: _KeopsSqrt.apply(((x[:, None, :] - y[None, :, :]) ** 2).sum(-1))            # noqa: E731
    else:
        raise NotImplementedError(formula)

    def log_conv(x, y, b, p, ranges=None):
        v = b.view(1, -1) - p * cost(x, y)
        if ranges is not None:
            v = torch.where(expand_ranges(ranges, len(x), len(y)), v, torch.full_like(v, -float("inf")))
        return v.logsumexp(1, keepdim=True)

    return log_conv


class Lazy:
    """Just enough of pykeops.torch.LazyTensor for `_legacy/utils.py:26-61` and `kernel_samples.py:62-146`: symbolic (N, M)
    arrays held densely in `.d`; `.ranges` (KeOps 6-tuple) restricts `@` to the kept blocks; `.t()` swaps axes and ranges."""

    def __init__(self, d, ranges=None):
        self.d, self.ranges = d, ranges

    def __sub__(self, o):
        return Lazy(self.d - o.d)

    def __pow__(self, k):
        return Lazy(self.d**k)

    def __neg__(self):
        return Lazy(-self.d)

    def __truediv__(self, k):
        return Lazy(self.d / k)

    def sum(self, dim):
        return Lazy(self.d.sum(dim))

    def exp(self):
        return Lazy(self.d.exp())

    def sqrt(self):
        return Lazy(self.d.clamp_min(1e-8).sqrt() if CLAMP_NORM2 else _KeopsSqrt.apply(self.d))

    def __matmul__(self, v):
        d = self.d
        if self.ranges is not None:
            d = d * expand_ranges(self.ranges, d.shape[-2], d.shape[-1]).to(d.dtype)
        return d @ v

    def t(self):
        return Lazy(self.d.transpose(-1, -2), None if self.ranges is None else swap_axes(self.ranges))


def _install():
    for mod in (ref_ss, ref_ks):
        mod.keops_available = True
        mod.grid_cluster, mod.cluster_ranges_centroids, mod.sort_clusters = grid_cluster, cluster_ranges_centroids, sort_clusters
        mod.from_matrix, mod.swap_axes = from_matrix, swap_axes
    ref_ss.generic_logsumexp = generic_logsumexp
    ref_utils.keops_available = True
    ref_utils.LazyTensor = Lazy
    ref_ks.LazyTensor = Lazy


_install()
from geomloss import SamplesLoss  # noqa: E402


# ---- cases -----------------------------------------------------------------------------------------------------------

def clouds(seed, N, M, D, same_law=False, weights=False, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, D, generator=g, dtype=torch.float64)
    y = torch.rand(M, D, generator=g, dtype=torch.float64)
    if not same_law:
        y = y * 0.7 + 0.2
    x, y = x + shift, y + shift
    if weights:
        a = torch.rand(N, generator=g, dtype=torch.float64) + 0.5
        b = torch.rand(M, generator=g, dtype=torch.float64) + 0.5
        a, b = a / a.sum(), b / b.sum()
    else:
        a, b = torch.full((N,), 1.0 / N, dtype=torch.float64), torch.full((M,), 1.0 / M, dtype=torch.float64)
    return a, x, b, y


SINKHORN = {
    # name: (SamplesLoss kwargs, cloud kwargs)
    "multiscale_p2_d3": (dict(p=2, blur=0.05, scaling=0.7), dict(seed=1, N=700, M=800, D=3)),
    "multiscale_p2_d3_w": (dict(p=2, blur=0.03, scaling=0.6), dict(seed=2, N=900, M=750, D=3, weights=True)),
    "multiscale_p1_d3": (dict(p=1, blur=0.02, scaling=0.7, truncate=2, cluster_scale=0.12), dict(seed=3, N=600, M=650, D=3)),
    "multiscale_p2_reach": (dict(p=2, blur=0.05, scaling=0.7, reach=0.3), dict(seed=4, N=640, M=600, D=3, weights=True)),
    "multiscale_p2_nodebias": (dict(p=2, blur=0.05, scaling=0.7, debias=False), dict(seed=5, N=600, M=700, D=3)),
    "multiscale_p2_truncate3": (dict(p=2, blur=0.04, scaling=0.8, truncate=3, cluster_scale=0.15), dict(seed=6, N=800, M=800, D=3)),
    "multiscale_p2_cluster_scale": (dict(p=2, blur=0.05, scaling=0.7, diameter=1.8, cluster_scale=0.1), dict(seed=7, N=700, M=640, D=3)),
    "multiscale_p2_last_jump": (dict(p=2, blur=0.1, scaling=0.6, cluster_scale=0.08), dict(seed=8, N=600, M=500, D=3)),
    "multiscale_p2_same_law": (dict(p=2, blur=0.05, scaling=0.7), dict(seed=9, N=800, M=800, D=3, same_law=True)),
    "multiscale_p2_d2_negative": (dict(p=2, blur=0.03, scaling=0.7, truncate=4, cluster_scale=0.08), dict(seed=10, N=900, M=1000, D=2, shift=-0.45)),
    "multiscale_p1_d1": (dict(p=1, blur=0.01, scaling=0.7, cluster_scale=0.03), dict(seed=11, N=500, M=400, D=1, weights=True)),
    "multiscale_p1_reach_d2": (dict(p=1, blur=0.03, scaling=0.7, reach=0.5, truncate=3, cluster_scale=0.1), dict(seed=12, N=600, M=600, D=2)),
}

KERNELS = {
    "kernel_multiscale_gaussian_d3": ("gaussian", dict(blur=0.05, truncate=5), dict(seed=21, N=900, M=800, D=3)),
    "kernel_multiscale_gaussian_d2_t3": ("gaussian", dict(blur=0.03, truncate=3), dict(seed=22, N=1000, M=900, D=2, weights=True)),
    "kernel_multiscale_gaussian_cluster_scale": ("gaussian", dict(blur=0.05, truncate=2, cluster_scale=1.5), dict(seed=23, N=700, M=700, D=3, shift=3.0)),
    "kernel_multiscale_gaussian_diameter": ("gaussian", dict(blur=0.04, truncate=4, diameter=2.0), dict(seed=24, N=800, M=600, D=3, weights=True)),
    "kernel_multiscale_laplacian_d3": ("laplacian", dict(blur=0.03, truncate=5), dict(seed=25, N=800, M=900, D=3)),
    "kernel_multiscale_laplacian_d1": ("laplacian", dict(blur=0.01, truncate=8), dict(seed=26, N=600, M=500, D=1, weights=True)),
    "kernel_multiscale_energy_d3": ("energy", dict(blur=0.05, truncate=5), dict(seed=27, N=500, M=600, D=3, weights=True)),
}


def run(loss_name, kw, a, x, b, y, dtype, rec, tag, grads=True):
    a, x, b, y = (t.to(dtype) for t in (a, x, b, y))
    L = SamplesLoss(loss_name, backend="multiscale", **kw)
    if grads:
        xg, ag = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
        val = L(ag, xg, b, y)
        gx, ga = torch.autograd.grad(val, [xg, ag], allow_unused=True)
        rec[f"loss_{tag}"] = val.detach().numpy()
        rec[f"gx_{tag}"] = gx.numpy()
        if ga is not None:
            rec[f"ga_{tag}"] = ga.numpy()
    else:
        rec[f"loss_{tag}"] = L(a, x, b, y).detach().numpy()
    F, G = SamplesLoss(loss_name, backend="multiscale", potentials=True, **kw)(a, x, b, y)
    rec[f"F_{tag}"], rec[f"G_{tag}"] = F.detach().numpy(), G.detach().numpy()
    if loss_name in ("gaussian", "laplacian") and kw.get("truncate") is not None:
        rec["perm_x"], rec["perm_y"] = PERMS[-2].numpy(), PERMS[-1].numpy()     # F[k] belongs to x[perm_x[k]]


def main():
    global CLAMP_NORM2
    only = set(sys.argv[1:])             # optional: regenerate the named cases only
    for name, (kw, ck) in SINKHORN.items():
        if only and name not in only:
            continue
        a, x, b, y = clouds(**ck)
        rec = dict(a=a.numpy(), x=x.numpy(), b=b.numpy(), y=y.numpy(), kwargs=np.array(repr(dict(loss="sinkhorn", **kw))))
        run("sinkhorn", kw, a, x, b, y, torch.float64, rec, "f64")
        run("sinkhorn", kw, a, x, b, y, torch.float32, rec, "f32")
        if kw["p"] == 1:
            CLAMP_NORM2 = False
            tmp = {}
            run("sinkhorn", kw, a, x, b, y, torch.float64, tmp, "f64")
            rec["loss_f64_keops_norm2"] = tmp["loss_f64"]
            CLAMP_NORM2 = True
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, rec["loss_f64"], rec["loss_f32"], rec.get("loss_f64_keops_norm2", ""))
    for name, (loss_name, kw, ck) in KERNELS.items():
        if only and name not in only:
            continue
        a, x, b, y = clouds(**ck)
        rec = dict(a=a.numpy(), x=x.numpy(), b=b.numpy(), y=y.numpy(), kwargs=np.array(repr(dict(loss=loss_name, **kw))))
        run(loss_name, kw, a, x, b, y, torch.float64, rec, "f64")
        run(loss_name, kw, a, x, b, y, torch.float32, rec, "f32")
        if loss_name != "gaussian":
            CLAMP_NORM2 = False
            tmp = {}
            run(loss_name, kw, a, x, b, y, torch.float64, tmp, "f64")
            rec["loss_f64_keops_norm2"] = tmp["loss_f64"]
            CLAMP_NORM2 = True
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, rec["loss_f64"], rec["loss_f32"], rec.get("loss_f64_keops_norm2", ""))


if __name__ == "__main__":
    main()
