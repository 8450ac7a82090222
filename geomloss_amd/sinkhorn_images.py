"""Sinkhorn divergence between measures on regular grids: images (B,K,N,N) and volumes (B,K,N,N,N).

Mirror of the reference's ``_legacy/sinkhorn_images.py``: the same symmetric, annealed loop as for
point clouds (``sinkhorn_divergence.sinkhorn_loop``), run coarse-to-fine over a dyadic pyramid of the two densities,
with the separable grid soft-min (``utils.softmin_grid`` -> HIP kernel ``glhip_lse_lines_fwd``) as the plugged-in
reduction.  Cost: |x-y|^p / p between pixel centres of the unit square / cube.
"""

import torch

from . import hip
from .sinkhorn_divergence import scaling_parameters, sinkhorn_cost, sinkhorn_loop
from .utils import log_dens, pyramid, softmin_grid, upsample

# The loop below is ~500 small launches whatever the image size (the temperatures do not depend on the data: the
# domain is the unit cube), and nothing in it is differentiable (every soft-min input is detached, as in the reference):
# with hipGraph mode on (GEOMLOSS_HIP_GRAPH=1 / sinkhorn_samples.set_graph_mode) it is captured once per
# (shape, parameters) and replayed as one graph launch.
_graphs = hip.GraphCache()


def extrapolate(f_ba, g_ab, eps, damping, C_xy, b_log, C_xy_fine):
    """Coarse-to-fine transfer of a potential: plain interpolation (``sinkhorn_images.py:7-8``)."""
    return upsample(f_ba)


def kernel_truncation(C_xy, C_yx, C_xy_fine, C_yx_fine, f_ba, g_ab, eps, truncate=None, cost=None, verbose=False):
    """Nothing to truncate on a grid: the separable kernel is dense but cheap (``:11-23``)."""
    return C_xy_fine, C_yx_fine


def sinkhorn_divergence(a, b, p=2, blur=None, reach=None, axes=None, scaling=0.5, cost=None, debias=True,
                        potentials=False, verbose=False, **kwargs):
    """Sinkhorn divergence (or the dual potentials) between two batches of densities on the same grid (``:26-202``).

    a, b: (B,K,N,N) or (B,K,N,N,N) non-negative tensors on a GPU, N a power of two.  ``blur`` defaults to one pixel,
    1/N.  Returns (B,) values, or two tensors shaped like a and b with ``potentials=True``.  As in the reference,
    ``axes`` is accepted and ignored (the domain is the unit cube) and custom costs are not implemented."""
    if blur is None:
        blur = 1 / a.shape[-1]
    if cost is not None:
        raise NotImplementedError()
    if scaling < 0.5:
        raise ValueError(f"Scaling value of {scaling} is too small: please use a number in [0.5, 1).")

    diameter, eps, eps_list, rho = scaling_parameters(None, None, p, blur, reach, 1, scaling)

    def solve(a, b):
        a_s, b_s = pyramid(a)[1:], pyramid(b)[1:]           # 2x2 ... NxN
        a_logs, b_logs = [log_dens(t) for t in a_s], [log_dens(t) for t in b_s]
        C_s = [p] * len(a_logs)                              # the "cost object" of a level is just the exponent

        # jump to the next finer level as soon as its pixels are resolved by the current temperature (``:153-161``)
        pixel = [diameter / t.shape[-1] for t in a_s]
        current, jumps = pixel.pop(0), []
        for i, e in enumerate(eps_list[1:]):
            if current**p > e:
                jumps.append(i + 1)
                current = pixel.pop(0)
        if verbose:
            print("Temperatures: ", eps_list)
            print("Jumps: ", jumps)
        assert len(jumps) == len(a_s) - 1, "There's a bug in the multicale pre-processing..."

        return sinkhorn_loop(
            softmin_grid, a_logs, b_logs, C_s, C_s, C_s, C_s, eps_list, rho,
            jumps=jumps, kernel_truncation=kernel_truncation, extrapolate=extrapolate, debias=debias,
        )

    from .sinkhorn_samples import graph_mode

    if graph_mode() and a.is_cuda and not verbose:
        key = (tuple(a.shape), a.dtype, a.device.index, p, float(blur), reach, float(scaling), debias)
        with torch.no_grad():
            f_aa, g_bb, g_ab, f_ba = _graphs.run(key, solve, (a, b))
        torch.autograd.set_grad_enabled(True)   # what sinkhorn_loop leaves behind (reference behaviour)
    else:
        f_aa, g_bb, g_ab, f_ba = solve(a, b)
    return sinkhorn_cost(eps, rho, a, b, f_aa, g_bb, g_ab, f_ba, batch=True, debias=debias, potentials=potentials)
