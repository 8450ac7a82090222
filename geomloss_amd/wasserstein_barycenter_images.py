"""Debiased Sinkhorn barycenters of images, on the HIP grid soft-min.

Same algorithm, arguments and result as ``ImagesBarycenter`` of the reference (``_legacy/wasserstein_barycenter_images.py:36-93``,
one iteration: ``:6-33``): K couplings between the input densities and the barycenter are refined together with the barycenter
and its debiasing density, on a pyramid of 2x2, 4x4, ..., NxN images, ``scaling_N`` iterations per level while the blur radius
shrinks from the image size to ``blur``.  Every soft-min is ``utils.softmin_grid`` (``glhip_lse_lines_fwd``, one pass per axis).
"""

import math

import torch

from .utils import log_dens, pyramid, softmin_grid, upsample


def _pseudo_barycenter(eps, p, ak_log, g_k, d_log, w):
    """log-density of the weighted geometric mean of the K measures pushed through their current couplings (``:7-11,24-27``)."""
    pushed = softmin_grid(eps, p, ak_log + g_k / eps) / eps          # (B,K,n,n)
    return d_log - (pushed * w).sum(1, keepdim=True)                 # (B,1,n,n)


def barycenter_iteration(f_k, g_k, d_log, eps, p, ak_log, w_k):
    """One symmetric update of the K couplings, of the barycenter and of its debiasing density (``:6-33``).
    Returns ``(f_k, g_k, d_log, bar_log)``."""
    w = w_k[:, :, None, None]
    bar_log = _pseudo_barycenter(eps, p, ak_log, g_k, d_log, w)
    towards_bar = softmin_grid(eps, p, ak_log + g_k / eps)           # measures -> barycenter
    towards_meas = softmin_grid(eps, p, bar_log + f_k / eps)         # barycenter -> measures
    f_k, g_k = (f_k + towards_bar) / 2, (g_k + towards_meas) / 2
    bar_log = _pseudo_barycenter(eps, p, ak_log, g_k, d_log, w)
    d_log = 0.5 * (d_log + bar_log + softmin_grid(eps, p, d_log) / eps)
    return f_k, g_k, d_log, bar_log


def _descent(n_levels, scaling_N, blur, p):
    """(pyramid level, temperature) of every iteration of the multiscale descent (``:66-80``): ``scaling_N`` iterations per level,
    the blur radius divided by 2^(1/scaling_N) after each — halved from one level to the next — and floored at ``blur``."""
    sigma = 1
    for level in range(n_levels):
        for _ in range(scaling_N):
            yield level, sigma**p
            sigma = max(sigma * (2 ** (-1 / scaling_N)), blur)


def ImagesBarycenter(measures, weights, blur=0, p=2, scaling_N=10, backward_iterations=5):
    """Barycenter (B,1,N,N) of the K densities ``measures`` (B,K,N,N) with barycentric weights ``weights`` (B,K).

    ``blur`` = 0 means one pixel.  Gradients with respect to ``measures`` / ``weights`` flow through ``backward_iterations`` extra
    iterations at the finest level (through the whole descent when it is 0), as in the reference."""
    if blur == 0:
        blur = 1 / measures.shape[-1]
    differentiable = measures.requires_grad or weights.requires_grad

    with torch.set_grad_enabled(backward_iterations == 0):
        levels = [log_dens(img) for img in pyramid(measures)[1:]]      # log-densities at 2x2, 4x4, ..., NxN (the 1x1 level is dropped)
        coarsest = levels[0]
        f_k = softmin_grid(1, p, coarsest)
        g_k = softmin_grid(1, p, coarsest)
        B, _, n0, n1 = coarsest.shape
        d_log = torch.full((B, 1, n0, n1), -math.log(n0 * n1), dtype=coarsest.dtype, device=coarsest.device)   # uniform density (:60-64)
        at, eps, bar_log = 0, 1, None
        for level, eps in _descent(len(levels), scaling_N, blur, p):
            if level != at:                                             # next pyramid level: refine the three maps (:82-85)
                f_k, g_k, d_log = upsample(f_k), upsample(g_k), upsample(d_log)
                at = level
            f_k, g_k, d_log, bar_log = barycenter_iteration(f_k, g_k, d_log, eps, p, levels[at], weights)

    if differentiable and backward_iterations > 0:
        for _ in range(backward_iterations):
            f_k, g_k, d_log, bar_log = barycenter_iteration(f_k, g_k, d_log, eps, p, levels[-1], weights)
    return bar_log.exp()
