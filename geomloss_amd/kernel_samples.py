"""Kernel norms (MMDs) between sampled measures: ``SamplesLoss("gaussian" | "laplacian" | "energy")``.

    Loss(a, b) = 1/2 <a, k*a> + 1/2 <b, k*b> - <a, k*b>,
    k(x,y) = exp(-|x-y|^2 / 2 blur^2) | exp(-|x-y| / blur) | -|x-y|.

Mirror of the reference's ``_legacy/kernel_samples.py``.  ``kernel_tensorized`` builds dense kernel
matrices with PyTorch (any device); ``kernel_online`` and ``kernel_multiscale`` evaluate the three
kernel-matrix x vector products with the HIP kernel ``glhip_kernel_conv_fwd`` (dense / block-sparse)
and never materialise a matrix.
"""

from functools import partial

import numpy as np
import torch

from . import hip
from .cluster import (block_ranges_device, cluster_ranges_centroids, clusterize_device, from_matrix, grid_cluster,
                      native_clustering_applies, sort_clusters)
from .utils import distances, scal, scal_sum, squared_distances


class DoubleGrad(torch.autograd.Function):
    """Identity whose gradient is doubled: lets the symmetric terms <a, K_xx a> be built with one detached side (``:43-54``)."""

    @staticmethod
    def forward(ctx, input):
        return input

    @staticmethod
    def backward(ctx, grad_output):
        return 2 * grad_output


def double_grad(x):
    return DoubleGrad.apply(x)


# ==============================================================================
#                          dense kernel matrices
# ==============================================================================


def gaussian_kernel(x, y, blur=0.05, **kwargs):
    return (-squared_distances(x / blur, y / blur) / 2).exp()


def laplacian_kernel(x, y, blur=0.05, **kwargs):
    return (-distances(x / blur, y / blur)).exp()


def energy_kernel(x, y, blur=None, **kwargs):
    return -distances(x, y)


kernel_routines = {
    "gaussian": gaussian_kernel,
    "laplacian": laplacian_kernel,
    "energy": energy_kernel,
}


class _LazyKernel:
    """Stand-in for the KeOps LazyTensor the reference builds at ``:62-82``: supports ``K @ v`` and ``K.t()``."""

    def __init__(self, name, x, y, blur, ranges=None, flags=0):
        self.name, self.x, self.y, self.blur, self.ranges, self.flags = name, x, y, blur, ranges, flags

    def __matmul__(self, v):  # v: (..., M, 1)
        return hip.kernel_conv(self.name, self.x, self.y, v.squeeze(-1), self.blur, self.ranges, flags=self.flags).unsqueeze(-1)

    def t(self):
        return _LazyKernel(self.name, self.y, self.x, self.blur, None if self.ranges is None else self.ranges.t(), self.flags)


def _matvec(K, v):
    """K @ v for a dense (..., N, M) matrix or a :class:`_LazyKernel`, with v of shape (..., M)."""
    return (K @ v.unsqueeze(-1)).squeeze(-1)


def _family_flags(x, y):
    """GRAD_FAMILY when one of the three products of the loss will run on a product-and-gradient kernel (see _kernel_operators)."""
    if (x.shape[-1] <= 3 and torch.is_grad_enabled() and hip.kernel_grad_fusion() and (x.requires_grad or y.requires_grad)):
        return hip.FLAG_GRAD_FAMILY
    return 0


def _kernel_operators(x, y, blur, kernel, name, lazy, ranges):
    """(K_xx, K_yy, K_xy).  Symmetric blocks differentiate through their first argument only, with a doubled
    gradient (kernel_samples.py:117-125)."""
    r_xx, r_yy, r_xy = ranges
    if lazy:
        if kernel is not None:
            raise NotImplementedError(
                "geomloss_amd: custom 'kernel' functions need dense matrices; use backend='tensorized'."
            )
        if name not in kernel_routines:
            raise KeyError(name)
        # A kernel norm is a difference of three large terms (two samples of one law: 1e-5 left of 1e-3).  When gradients are on,
        # the products whose first cloud requires them run on the product-and-gradient kernel (hip._KernelConv); the other
        # products of the same loss are then sent to the kernel of the same family (gaussian: 16x16x32 MFMA tiling, identical
        # exponent arithmetic; laplacian / energy: explicit differences with |.| = m rsq(m), bit-identical to the product of
        # the fused mode) so that the per-term rounding bias stays common to the three terms and cancels.
        flags = _family_flags(x, y)
        build = lambda u, w, r: _LazyKernel(name, u, w, blur, r, flags)  # noqa: E731
    else:
        dense = kernel_routines[name] if kernel is None else kernel
        build = lambda u, w, r: dense(u, w, blur=blur)  # noqa: E731
    return build(double_grad(x), x.detach(), r_xx), build(double_grad(y), y.detach(), r_yy), build(x, y, r_xy)


# The matrix-free dense path evaluates the norm on the UNION cloud.  With z = (x, y) and the signed weights w = (α, -β),
#
#     Loss = 1/2 <α-β, k*(α-β)> = 1/2 <w, K_zz w> = 1/2 ( <α, U_x> - <β, U_y> ),   U = K_zz w  (rows of x, rows of y)
#
# and U_x = k*α - k*β on x, -U_y = k*β - k*α on y are the potentials of ``:139-141``.  It is the same number as the reference's
# three terms 1/2 <α, K_xx α> + 1/2 <β, K_yy β> - <α, K_xy β> (``:143-146``), but those are three LARGE terms: between two samples
# of one law at 1e6 points the gaussian loss is 5e-4 of each of them, the energy distance 6e-7 — and every fast kernel here carries
# a one-sided rounding bias of 1e-6 ... 1e-5 per kernel value (the float32 accumulation of the bf16x3 exponent inside the MFMA,
# DESIGN §4.3c), which the three-term form only survives if the three launches happen to share it (round 3: 4e-4 off at 1e6
# without gradients, 2.4e-4 between the gradient and the no-gradient answer of one input).  In the union form the positive and
# negative columns of a row are summed by ONE launch, relative to ONE centre: whatever bias a kernel value carries multiplies
# a difference, not a term, and the relative error of the loss is that of a kernel value.  The dot products are carried in
# float64 (utils.scal on GPU tensors).  Pair evaluations: (N + M)^2 instead of N^2 + M^2 + N M; the value-only case below
# halves that again.
#
# Value only (no gradient flows, no potentials), one big un-batched problem: <w, K_zz w> is a symmetric quadratic form, so the
# pairs (i, j) and (j, i) need not both be evaluated.  The union cloud is put in compact order (hip.compact_order: x and y points
# interleaved along a voxel path, every 256-row block compact), cut into blocks of 256 rows (the row tile of the kernels); block
# I reduces once over its own columns (the diagonal block, both orientations inside it) and once over the columns of the blocks
# after it, counted twice — two block-sparse launches of the ordinary product kernel, (N + M)^2 / 2 pair evaluations.
_UPPER_BLOCK, _UPPER_CHUNKS, _UPPER_MIN_PAIRS = 256, 8, 2e9
_UPPER_KERNELS = ("gaussian", "laplacian", "energy")


def _upper_triangle_patterns(N, device):
    """(diagonal, strictly upper) block-sparse patterns of an N x N product in row blocks of 256: KeOps-style ranges."""
    R, nc = _UPPER_BLOCK, _UPPER_CHUNKS
    C = (N + R - 1) // R
    first = torch.arange(C, device=device, dtype=torch.int32) * R
    rows = torch.stack((first, (first + R).clamp_max(N)), 1).contiguous()
    one = torch.arange(1, C + 1, device=device, dtype=torch.int32)
    diag = hip.BlockRanges(rows, one.contiguous(), rows.clone(), None, None, None)
    # block I -> columns [end of block I, N), in `nc` pieces (32-aligned) so that the column splits of the launch share the long rows
    lo = rows[:, 1].long()
    step = ((N - lo + nc - 1) // nc + 31) // 32 * 32                                 # (C,)
    k = torch.arange(nc, device=device).view(1, -1)
    js = (lo.view(-1, 1) + k * step.view(-1, 1)).clamp_max(N)
    je = (js + step.view(-1, 1)).clamp_max(N)
    red = torch.stack((js, je), 2).view(-1, 2).int().contiguous()                    # empty pieces (js == je) are skipped by the kernels
    upper = hip.BlockRanges(rows, (one * nc).contiguous(), red, None, None, None)
    return diag, upper


def _quadratic_form_value(name, z, w, blur):
    """<w, K_zz w> for one un-batched cloud z (1,N,D)|(N,D) with (signed) weights w, from the upper triangle; no autograd graph;
    a 0-dim float64 tensor.  The cloud is first put in the compact order of hip.compact_order: the 256-row blocks are then what
    the matrix-core distance kernels want (hip._KernelConv switches them on for block-sparse launches), and the points of the two
    measures of a norm are interleaved in every row block and every column tile."""
    zd = z.detach().reshape(-1, z.shape[-1])
    wv = w.detach().reshape(-1).float()
    perm, zd = hip.compact_order(zd)
    wv = wv[perm]
    diag, upper = _upper_triangle_patterns(zd.shape[0], zd.device)
    d = hip.kernel_conv(name, zd, zd, wv, blur, ranges=diag)
    u = hip.kernel_conv(name, zd, zd, wv, blur, ranges=upper)
    return torch.dot(wv.double(), d.double() + 2.0 * u.double())


def _takes_no_gradient(*tensors):
    return not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors))


def _upper_triangle_applies(name, pts):
    """One un-batched cloud of dimension <= 3, big enough for the two block-sparse launches to pay."""
    return (name in _UPPER_KERNELS and pts.shape[-1] <= 3 and float(pts.shape[-2]) ** 2 >= _UPPER_MIN_PAIRS
            and (pts.dim() == 2 or pts.shape[0] == 1) and pts.dtype != torch.float64)      # (the compact order is an fp32 / bf16 kernel)


def _weights_dtype(points):
    """Weights, products and sums are fp32 next to fp32 / bf16 / fp16 clouds, fp64 next to fp64 ones."""
    return torch.float64 if points.dtype == torch.float64 else torch.float32


class _UnionNorm(torch.autograd.Function):
    """1/2 <w, K_zz w> on the union cloud z = (x, y), w = (α, -β), as ONE autograd node.

    forward: the two row passes U_x = (K_zz w)|x, U_y = (K_zz w)|y, each together with d U_i / d z_i when its cloud takes a gradient
    (one product-and-gradient reduction).  backward, first order: elementwise — d/dx_i = α_i d U_i / d x_i (the quadratic form is
    symmetric: the derivative through the columns equals the one through the rows, which is what the reference's DoubleGrad
    trick expresses, ``_legacy/kernel_samples.py:43-54,117-125``), d/dα = U_x, d/dβ = -U_y.
    backward under ``create_graph=True`` (gaussian kernel): the same gradients written with differentiable kernel products of the
    NON-detached clouds and weights, so that second derivatives are those of the loss itself — including the dependence through
    the columns that a detached copy drops (the reference's Hessian of a self-term is incomplete for that reason)."""

    @staticmethod
    def forward(ctx, name, blur, α, x, β, y):
        batch = x.dim() > 2
        z = torch.cat((x, y.to(x.dtype)), dim=-2)
        w = torch.cat((α.to(_weights_dtype(x)), -β.to(_weights_dtype(x))), dim=-1)
        U_x, unit_x = hip.kernel_conv_with_unit(name, x, z, w, blur, ctx.needs_input_grad[3])
        U_y, unit_y = hip.kernel_conv_with_unit(name, y, z, w, blur, ctx.needs_input_grad[5])
        ctx.name, ctx.blur, ctx.units = name, blur, (unit_x, unit_y)
        ctx.save_for_backward(α, x, β, y, U_x, U_y)
        return 0.5 * scal_sum(α, U_x, β, -U_y, batch=batch)

    @staticmethod
    def backward(ctx, gL):
        if torch.is_grad_enabled() and ctx.name == "gaussian":
            return _UnionNorm._differentiable_backward(ctx, gL)
        return _UnionNorm._backward_once(ctx, gL)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def _backward_once(ctx, gL):
        α, x, β, y, U_x, U_y = ctx.saved_tensors
        name, blur = ctx.name, ctx.blur
        gl = gL.reshape(-1, 1) if x.dim() > 2 else gL.reshape(())           # one factor per batch item
        gα = gβ = gx = gy = None
        if ctx.needs_input_grad[2]:
            gα = (gl * U_x).to(α.dtype).reshape(α.shape)
        if ctx.needs_input_grad[4]:
            gβ = (-gl * U_y).to(β.dtype).reshape(β.shape)
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[5]:
            z = torch.cat((x, y.to(x.dtype)), dim=-2)
            w = torch.cat((α.to(_weights_dtype(x)), -β.to(_weights_dtype(x))), dim=-1)

            def rows(pts, wt, unit):     # d/d pts_i of 1/2 <w, K w> = wt_i d U_i / d pts_i
                g = gl * wt.to(w.dtype)
                if unit is not None:
                    return (g.unsqueeze(-1) * unit).to(pts.dtype).reshape(pts.shape)
                return hip.kernel_conv_row_gradient(name, pts, z, w, g, blur).to(pts.dtype)
            if ctx.needs_input_grad[3]:
                gx = rows(x, α, ctx.units[0])
            if ctx.needs_input_grad[5]:
                gy = rows(y, -β, ctx.units[1])
        return None, None, gα, gx, gβ, gy

    @staticmethod
    def _differentiable_backward(ctx, gL):
        α, x, β, y = ctx.saved_tensors[:4]
        name, blur = ctx.name, ctx.blur
        batch = x.dim() > 2
        gl = gL.reshape(-1, 1) if batch else gL.reshape(())
        s = 1.0 / (blur * blur)
        z = torch.cat((x, y.to(x.dtype)), dim=-2)
        w = torch.cat((α, -β), dim=-1).to(_weights_dtype(x))      # fp32 next to bf16 / fp16 clouds, like forward and _backward_once
        D = x.shape[-1]

        def rows(pts, wt, U):            # wt_i sum_j w_j grad_1 k(pts_i, z_j) = -s wt_i [ pts_i U_i - (K (w z))_i ]
            Kwz = torch.stack([hip.kernel_conv(name, pts, z, w * z[..., d].to(w.dtype), blur) for d in range(D)], dim=-1)
            return (gl * wt.to(w.dtype)).unsqueeze(-1) * (-s) * (pts.to(w.dtype) * U.unsqueeze(-1) - Kwz)
        U_x = hip.kernel_conv(name, x, z, w, blur)
        U_y = hip.kernel_conv(name, y, z, w, blur)
        gα = gl * U_x if ctx.needs_input_grad[2] else None
        gβ = -gl * U_y if ctx.needs_input_grad[4] else None
        gx = rows(x, α, U_x) if ctx.needs_input_grad[3] else None
        gy = rows(y, -β, U_y) if ctx.needs_input_grad[5] else None
        return None, None, gα, gx, gβ, gy


def _kernel_loss_union(α, x, β, y, blur, name, potentials):
    """The matrix-free dense kernel norm (or its potentials) on the union cloud: see the note above."""
    if name not in kernel_routines:
        raise KeyError(name)
    batch = x.dim() > 2
    z = torch.cat((x.detach(), y.detach().to(x.dtype)), dim=-2)
    w = torch.cat((α.detach().to(_weights_dtype(x)), -β.detach().to(_weights_dtype(x))), dim=-1)

    if not potentials and _takes_no_gradient(α, x, β, y) and _upper_triangle_applies(name, z):
        out = (0.5 * _quadratic_form_value(name, z, w, blur)).float()
        return out.view(1) if batch else out

    if not potentials:
        if _takes_no_gradient(α, x, β, y):
            with torch.no_grad():
                U_x, U_y = hip.kernel_conv(name, x, z, w, blur), hip.kernel_conv(name, y, z, w, blur)
                return 0.5 * scal_sum(α, U_x, β, -U_y, batch=batch)
        return _UnionNorm.apply(name, 1.0 if blur is None else float(blur), α, x, β, y)

    # potentials (``:139-141``) that nobody differentiates (kernel_loss sends the others to the four products of the reference):
    # rows of x, rows of y of the same signed product
    with torch.no_grad():
        U_x = hip.kernel_conv(name, x, z, w, blur)      # (k*α - k*β)(x_i)
        U_y = hip.kernel_conv(name, y, z, w, blur)      # (k*α - k*β)(y_j)
    return U_x, -U_y


def kernel_loss(
    α, x, β, y, blur=0.05, kernel=None, name=None, potentials=False, use_keops=False,
    ranges_xx=None, ranges_yy=None, ranges_xy=None, **kwargs,
):
    """Kernel norm 1/2 <α-β, k*(α-β)> or its potentials (``:92-146``).  ``use_keops=True`` selects the matrix-free
    HIP path (the keyword keeps the reference's name; no KeOps is involved)."""
    batch = x.dim() > 2
    if use_keops and kernel is None and ranges_xx is None and ranges_yy is None and ranges_xy is None:
        # Potentials that somebody differentiates keep the reference's four products (below): F = K(2x, x̄) ᾱ - K(x, y) β is
        # differentiable once in x through the self term and in x, y AND β through the cross term (``:117-141``), a structure the
        # union form — all columns and weights constants — cannot reproduce (round-4 advice: ∂F/∂y, ∂F/∂β, ∂G/∂x, ∂G/∂α came out zero).
        if not potentials or _takes_no_gradient(α, x, β, y):
            return _kernel_loss_union(α, x, β, y, blur, name, potentials)

    K_xx, K_yy, K_xy = _kernel_operators(x, y, blur, kernel, name, use_keops, (ranges_xx, ranges_yy, ranges_xy))

    a_x = _matvec(K_xx, α.detach())  # (k * α)(x_i)
    b_y = _matvec(K_yy, β.detach())  # (k * β)(y_j)
    b_x = _matvec(K_xy, β)           # (k * β)(x_i)

    if potentials:
        K_yx = K_xy.t() if use_keops else K_xy.transpose(-1, -2)
        return a_x - b_x, b_y - _matvec(K_yx, α)

    self_terms = scal(double_grad(α), a_x, batch=batch) + scal(double_grad(β), b_y, batch=batch)
    return 0.5 * self_terms - scal(α, b_x, batch=batch)


kernel_tensorized = partial(kernel_loss, use_keops=False)
kernel_online = partial(kernel_loss, use_keops=True)


def max_diameter(x, y):
    mins = torch.minimum(x.min(dim=0)[0], y.min(dim=0)[0])
    maxs = torch.maximum(x.max(dim=0)[0], y.max(dim=0)[0])
    return (maxs - mins).norm().item()


def kernel_multiscale(
    α, x, β, y, blur=0.05, kernel=None, name=None, truncate=5, diameter=None, cluster_scale=None,
    potentials=False, verbose=False, **kwargs,
):
    """Block-sparse kernel norm: cluster pairs farther apart than (truncate + cell diameter) blurs are skipped (``:177-271``).

    Works on single, un-batched measures.  As in the reference, the clouds are centred and sorted by
    cluster, and returned potentials follow the sorted order.
    """
    if truncate is None or name == "energy":
        return kernel_online(
            α.unsqueeze(0), x.unsqueeze(0), β.unsqueeze(0), y.unsqueeze(0),
            blur=blur, kernel=kernel, truncate=truncate, name=name, potentials=potentials, **kwargs,
        )

    # centre and rescale: truncation thresholds are expressed in units of blur
    center = (x.mean(-2, keepdim=True) + y.mean(-2, keepdim=True)) / 2
    x, y = x - center, y - center
    x_, y_ = x / blur, y / blur
    D = x.shape[-1]

    if cluster_scale is None:
        diameter = max_diameter(x_.view(-1, D), y_.view(-1, D)) if diameter is None else diameter / blur
        cluster_scale = diameter / (np.sqrt(D) * 2000 ** (1 / D))
    cell_diameter = cluster_scale * np.sqrt(D)

    if native_clustering_applies(x) and native_clustering_applies(y) and α.dim() == 1:
        # device path: glhip_grid_cluster clusters x / blur itself (pre_div), glhip_block_ranges applies the geometric rule
        _, α, x_c, x, ranges_x, _ = clusterize_device(α, x, cluster_scale, pre_div=blur)
        _, β, y_c, y, ranges_y, _ = clusterize_device(β, y, cluster_scale, pre_div=blur)
        if verbose:
            print("{}x{} clusters, computed at scale = {:2.3f}".format(len(x_c), len(y_c), cluster_scale))
        reach2 = (truncate + cell_diameter) ** 2
        ranges_xx = block_ranges_device("within", x_c, x_c, None, None, ranges_x, ranges_x, reach2)
        ranges_yy = block_ranges_device("within", y_c, y_c, None, None, ranges_y, ranges_y, reach2)
        ranges_xy = block_ranges_device("within", x_c, y_c, None, None, ranges_x, ranges_y, reach2)
        return kernel_loss(
            α, x, β, y, blur=blur, kernel=kernel, name=name, potentials=potentials, use_keops=True,
            ranges_xx=ranges_xx, ranges_yy=ranges_yy, ranges_xy=ranges_xy,
        )

    x_lab = grid_cluster(x_, cluster_scale)
    y_lab = grid_cluster(y_, cluster_scale)
    ranges_x, x_c, _ = cluster_ranges_centroids(x_, x_lab, weights=α)
    ranges_y, y_c, _ = cluster_ranges_centroids(y_, y_lab, weights=β)

    if verbose:
        print("{}x{} clusters, computed at scale = {:2.3f}".format(len(x_c), len(y_c), cluster_scale))

    (α, x), x_lab = sort_clusters((α, x), x_lab)
    (β, y), y_lab = sort_clusters((β, y), y_lab)

    with torch.no_grad():
        reach2 = (truncate + cell_diameter) ** 2
        ranges_xx = from_matrix(ranges_x, ranges_x, squared_distances(x_c, x_c) <= reach2)
        ranges_yy = from_matrix(ranges_y, ranges_y, squared_distances(y_c, y_c) <= reach2)
        ranges_xy = from_matrix(ranges_x, ranges_y, squared_distances(x_c, y_c) <= reach2)

    return kernel_loss(
        α, x, β, y, blur=blur, kernel=kernel, name=name, potentials=potentials, use_keops=True,
        ranges_xx=ranges_xx, ranges_yy=ranges_yy, ranges_xy=ranges_xy,
    )
