"""Debiased Sinkhorn barycenters of images (mirror of ``_legacy/wasserstein_barycenter_images.py``), on the HIP grid soft-min."""

import torch

from .utils import log_dens, pyramid, softmin_grid as softmin, upsample


def barycenter_iteration(f_k, g_k, d_log, eps, p, ak_log, w_k):
    """One symmetric update of the K couplings, of the barycenter and of its debiasing measure (``:6-33``)."""
    w = w_k[:, :, None, None]

    def barycenter_of(g):   # weighted geometric mean of the K "pseudo-steps" from the measures
        ft = softmin(eps, p, ak_log + g / eps) / eps                 # (B,K,n,n)
        return d_log - (ft * w).sum(1, keepdim=True)                 # (B,1,n,n)

    bar_log = barycenter_of(g_k)
    ft_k = softmin(eps, p, ak_log + g_k / eps)        # measures -> barycenter
    gt_k = softmin(eps, p, bar_log + f_k / eps)       # barycenter -> measures
    f_k, g_k = (f_k + ft_k) / 2, (g_k + gt_k) / 2
    bar_log = barycenter_of(g_k)
    d_log = 0.5 * (d_log + bar_log + softmin(eps, p, d_log) / eps)
    return f_k, g_k, d_log, bar_log


def ImagesBarycenter(measures, weights, blur=0, p=2, scaling_N=10, backward_iterations=5):
    """Barycenter (B,1,N,N) of K densities (B,K,N,N) with weights (B,K) (``:36-93``): multiscale descent with
    ``scaling_N`` iterations per pyramid level, blur halved from level to level down to ``blur`` (default: one pixel);
    gradients flow through ``backward_iterations`` extra iterations at the finest level."""
    a_k, w_k = measures, weights
    if blur == 0:
        blur = 1 / measures.shape[-1]

    with torch.set_grad_enabled(backward_iterations == 0):
        ak_s = pyramid(a_k)[1:]
        ak_log_s = [log_dens(t) for t in ak_s]

        sigma = 1
        eps = sigma**p
        f_k, g_k = softmin(eps, p, ak_log_s[0]), softmin(eps, p, ak_log_s[0])
        d_log = torch.ones_like(ak_log_s[0]).sum(dim=1, keepdim=True)
        d_log = d_log - d_log.logsumexp([2, 3], keepdim=True)

        for n, ak_log in enumerate(ak_log_s):
            for _ in range(scaling_N):
                eps = sigma**p
                f_k, g_k, d_log, bar_log = barycenter_iteration(f_k, g_k, d_log, eps, p, ak_log, w_k)
                sigma = max(sigma * (2 ** (-1 / scaling_N)), blur)
            if n + 1 < len(ak_s):
                f_k, g_k, d_log = upsample(f_k), upsample(g_k), upsample(d_log)

    if (measures.requires_grad or weights.requires_grad) and backward_iterations > 0:
        for _ in range(backward_iterations):
            f_k, g_k, d_log, bar_log = barycenter_iteration(f_k, g_k, d_log, eps, p, ak_log, w_k)

    return bar_log.exp()
