"""Dense helpers shared by the tensorized code paths (reference: ``_legacy/utils.py:13-61``) and the grid helpers of
the image / volume path (``_legacy/utils.py:64-279``)."""

import numpy as np
import torch
from torch.nn.functional import avg_pool2d, avg_pool3d, interpolate


def _dot(a, f, batch):
    if batch:
        B = a.shape[0]
        return (a.reshape(B, -1) * f.reshape(B, -1)).sum(1)
    return torch.dot(a.reshape(-1), f.reshape(-1))


_WIDEN_MIN = 16384      # points per measure from which the float64 accumulation is worth its ~8 extra (tiny) launches


def _widen(*tensors):
    """Big GPU vectors in single / half precision: dot products are carried in float64 and rounded once at the end.  (Small ones:
    a float32 dot of 1e4 terms is good to 1e-7, and a 1000-point loss is launch-bound — the conversions would cost 15 % of it.)"""
    return all(t.is_cuda and t.is_floating_point() and t.dtype != torch.float64 for t in tensors) and tensors[0].shape[-1] >= _WIDEN_MIN


def scal(a, f, batch=False):
    """<a, f>: one dot product, or one per batch item (``utils.py:13-18``).

    CPU tensors: the reference's expression, bit for bit.  GPU tensors in single or half precision: products and sum are carried
    in float64 and the result is rounded once (a float32 dot of 1e6 terms is good to ~1e-6 of its largest partial sum, and the
    losses between close 1e6-point measures are 1e-4 ... 1e-6 of theirs)."""
    if _widen(a, f):
        return _dot(a.double(), f.double(), batch).to(torch.result_type(a, f))
    return _dot(a, f, batch)


def scal_sum(a, f, b, g, batch=False):
    """<a, f> + <b, g>, the closing expression of every loss formula (``sinkhorn_divergence.py:171-250``).  CPU: the two float32
    dots added, as the reference does; GPU: both dots and their sum in float64, rounded once — the two halves of a Sinkhorn
    divergence between close measures cancel to 1e-4 of their size."""
    if _widen(a, f, b, g):
        out = _dot(a.double(), f.double(), batch) + _dot(b.double(), g.double(), batch)
        return out.to(torch.promote_types(torch.result_type(a, f), torch.result_type(b, g)))
    return scal(a, f, batch=batch) + scal(b, g, batch=batch)


def squared_distances(x, y):
    """Dense |x_i - y_j|^2 as |x|^2 - 2 x.y + |y|^2, for (N,D)x(M,D) or (B,N,D)x(B,M,D) (``utils.py:39-53``)."""
    if x.dim() not in (2, 3):
        print("x.shape : ", x.shape)
        raise ValueError("Incorrect number of dimensions")
    x2 = (x * x).sum(-1).unsqueeze(-1)  # (..., N, 1)
    y2 = (y * y).sum(-1).unsqueeze(-2)  # (..., 1, M)
    xy = torch.matmul(x, y.transpose(-1, -2))  # (..., N, M)
    return x2 - 2 * xy + y2


def distances(x, y):
    """Dense |x_i - y_j|, squared distances clamped at 1e-8 before the root (``utils.py:56-61``)."""
    return torch.sqrt(torch.clamp_min(squared_distances(x, y), 1e-8))


# ==============================================================================
#                     measures on regular grids: images (B,K,N,N), volumes (B,K,N,N,N)
# ==============================================================================

BATCH, CHANNEL, HEIGHT, WIDTH, DEPTH = 0, 1, 2, 3, 4


def dimension(I):
    """2 for images (B,K,N,N), 3 for volumes (B,K,N,N,N) (``utils.py:71-73``)."""
    return I.dim() - 2


def _subsample(I):
    """Mass-preserving 2x coarsening: sums over 2^D blocks (``utils.py:76-79``)."""
    return 4 * avg_pool2d(I, 2) if dimension(I) == 2 else 8 * avg_pool3d(I, 2)


def pyramid(I):
    """[1x1, 2x2, ..., NxN] coarsenings of a density, coarsest first (``utils.py:87-96``)."""
    levels = [I]
    for _ in range(int(np.log2(I.shape[HEIGHT]))):
        I = _subsample(I)
        levels.append(I)
    return levels[::-1]


def upsample(I):
    """2x bi- / tri-linear refinement of a dual potential (``utils.py:99-101``)."""
    mode = "bilinear" if dimension(I) == 2 else "trilinear"
    return interpolate(I, scale_factor=2, mode=mode, align_corners=False)


def log_dens(α):
    """log of a density, -10000 where it vanishes (``utils.py:104-107``)."""
    return torch.where(α > 0, α.log(), torch.full_like(α, -10000.0))     # no boolean-mask indexing (host round trip)


def C_transform(G, tau=1, p=2):
    """Hard C-transform of an array on a regular grid with unit pixels (``utils.py:116-182``):
    ``F(x_i) = max_j [G(x_j) - |x_i - x_j|^p / (p tau)]`` for G of shape (B,N), (B,N,N) or (B,N,N,N), one separable pass of the
    HIP kernel ``glhip_max_lines_fwd`` per axis.  Like the reference — whose ``if p == 1 / if p == 2 / else`` chain sends p = 1
    to the ``else`` — only p = 2 is accepted."""
    import numpy as np

    from . import hip

    if p != 2:
        raise NotImplementedError()
    D = G.ndim - 1
    if D not in (1, 2, 3):
        raise ValueError("C_transform expects (B,N), (B,N,N) or (B,N,N,N) arrays.")
    step = 1.0 / np.sqrt(2 * tau)                         # x = arange(N) / sqrt(2 tau), cost (x_i - x_j)^2
    last = G.dim() - 1
    out = hip.max_lines(G, step, p)
    for axis in range(last - 1, 0, -1):
        out = hip.max_lines(out.transpose(axis, last), step, p).transpose(axis, last)
    return out


def softmin_grid(eps, C_xy, h_y):
    """Soft-C-transform on a grid, one separable log-sum-exp pass per axis (``utils.py:190-279``).

    ``C_xy`` is the exponent p of the cost |x-y|^p / p between pixel centres i/N of the unit square / cube; ``h_y`` is
    (B,K,N,N) or (B,K,N,N,N).  Each pass is the HIP kernel ``glhip_lse_lines_fwd`` on the lines of one axis (the axis is
    moved to the last position and made contiguous, as the reference does for KeOps); differentiable in ``h_y``."""
    from . import hip

    D = dimension(h_y)
    p = C_xy
    if p not in (1, 2):
        raise NotImplementedError()
    if D not in (2, 3):
        raise ValueError("softmin_grid expects (B,K,N,N) images or (B,K,N,N,N) volumes.")
    last = h_y.dim() - 1
    out = hip.lse_lines(h_y, eps, p)                     # lines of the last axis
    for axis in range(last - 1, 1, -1):                  # then every other spatial axis, swapped into last position
        out = hip.lse_lines(out.transpose(axis, last), eps, p).transpose(axis, last)
    return -eps * out
