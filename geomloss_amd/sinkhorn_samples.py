"""Sinkhorn divergences between sampled measures: the three drivers behind ``SamplesLoss("sinkhorn")``.

Mirror of the reference's ``_legacy/sinkhorn_samples.py`` (same function names and arguments):

* ``sinkhorn_tensorized``  (``:74-221``)  dense (B,N,M) cost matrices.  CPU tensors: plain PyTorch
  (this is the reference algorithm, BASELINE config 1).  GPU tensors: the row-wise soft-min is the
  HIP kernel ``glhip_softmin_dense_fwd``.
* ``sinkhorn_online``      (``:349-424``) costs recomputed on the fly from the points — HIP only.
* ``sinkhorn_multiscale``  (``:547-681``) two-scale scheme with voxel clusters and kernel
  truncation — HIP only, block-sparse kernels.

The Sinkhorn loop itself lives in :mod:`geomloss_amd.sinkhorn_divergence`; the kernels are reached
through :mod:`geomloss_amd.hip`.
"""

import math
import os
from functools import partial

import numpy as np
import torch

from . import hip
from .cluster import (block_ranges_device, cluster_ranges_centroids, clusterize_device, clusterize_device_many, from_matrix,
                      grid_cluster, kept_pairs_device, native_clustering_applies, native_keep_rule_applies, swap_axes)
from .sinkhorn_divergence import log_weights, log_weights_many, max_diameter, scaling_parameters, sinkhorn_cost, sinkhorn_loop
from .utils import distances, squared_distances

# ==============================================================================
#                          backend == "tensorized"
# ==============================================================================

def _cost_p1(x, y):
    return distances(x, y)


def _cost_p2(x, y):
    return squared_distances(x, y) / 2


_cost_p1.glhip_exponent, _cost_p2.glhip_exponent = 1, 2     # lets kernel_truncation evaluate the keep rule on the device
cost_routines = {1: _cost_p1, 2: _cost_p2}


def softmin_tensorized(eps, C_xy, h_y):
    """``-eps * logsumexp_j(h_j - C_ij / eps)`` for a dense (B,N,M) cost matrix (``:32-71``)."""
    B = C_xy.shape[0]
    # the HIP row-reduction computes in fp32: double-precision matrices (and builds without the extension) stay on PyTorch
    if C_xy.is_cuda and C_xy.dtype != torch.float64 and hip.library_available():
        return hip.softmin_dense(eps, C_xy, h_y.view(B, -1)).to(C_xy.dtype)
    return -eps * (h_y.view(B, 1, -1) - C_xy / eps).logsumexp(2).view(B, -1)


def sinkhorn_tensorized(
    a, x, b, y, p=2, blur=0.05, reach=None, diameter=None, scaling=0.5, cost=None, debias=True,
    potentials=False, **kwargs,
):
    """Sinkhorn divergence on explicit cost matrices; a (B,N), x (B,N,D), b (B,M), y (B,M,D)."""
    if cost is None:
        cost = cost_routines[p]

    # right-hand sides are detached: gradients flow through the first argument of each cost only
    C_xy = cost(x, y.detach())
    C_yx = cost(y, x.detach())
    C_xx = cost(x, x.detach()) if debias else None
    C_yy = cost(y, y.detach()) if debias else None

    diameter, eps, eps_list, rho = scaling_parameters(x, y, p, blur, reach, diameter, scaling)

    f_aa, g_bb, g_ab, f_ba = sinkhorn_loop(
        softmin_tensorized, log_weights(a), log_weights(b), C_xx, C_yy, C_xy, C_yx, eps_list, rho, debias=debias
    )
    return sinkhorn_cost(eps, rho, a, b, f_aa, g_bb, g_ab, f_ba, batch=True, debias=debias, potentials=potentials)


# ==============================================================================
#                          backend == "online"
# ==============================================================================

# The reference describes the ground cost of its KeOps backends with formula strings; the HIP kernels
# implement exactly these two.
cost_formulas = {
    1: "Norm2(X-Y)",
    2: "(SqDist(X,Y) / IntCst(2))",
}


def _exponent_of(cost_formula):
    """Maps a cost formula of the reference's online / multiscale API to the kernel's ``p``."""
    for p, formula in cost_formulas.items():
        if isinstance(cost_formula, str) and cost_formula.replace(" ", "") == formula.replace(" ", ""):
            return p
    raise NotImplementedError(
        "geomloss_amd: the HIP backends implement the cost formulas "
        f"{list(cost_formulas.values())} (p = 1, 2); got {cost_formula!r}. "
        "Arbitrary cost functions are available with backend='tensorized'."
    )


def _fp32_weights(a, b):
    """bf16 / fp16 clouds are read as such by the kernels, but weights, dual potentials and the loss stay fp32."""
    if a.dtype in (torch.bfloat16, torch.float16):
        a, b = a.float(), b.float()
    return a, b


def softmin_online(eps, C_xy, h_y, p=2, flags=0):
    """Soft-C-transform on implicit costs (``:337-346`` and ``:229-290``): C_xy = (x, y), batched or not."""
    x, y = C_xy
    out = hip.softmin(eps, x, y, h_y, p=p, flags=flags)
    return out if x.dim() > 2 else out.view(1, -1)


class _HipSoftmin:
    """The soft-min callable handed to ``sinkhorn_loop`` by the HIP drivers: ``softmin(eps, C, h)`` as the reference
    defines it, plus the fused half-step ``step(...)`` that the loop uses where autograd is off (D <= 3)."""

    def __init__(self, p, multiscale):
        self.p, self.multiscale = p, multiscale
        self.h2_min_eps = float("inf")      # see set_range: temperatures from which the f16 x 2 exponent layout is in range
        self._plan = None   # (x, y, a_log, b_log, debias, hip.Iter4Plan) of the loop being run

    def set_range(self, extent):
        """Tells the soft-min how wide the clouds ARE (the diagonal of their bounding box, or an upper bound of it; None: unknown),
        so that it can ask for the f16 x 2 exponent layout (GLHIP_FLAG_F16X2: about half the matrix instructions and LDS bytes of
        the default bf16 x 3 one) where the exponents fit f16's range: terms of size log2(e) extent^2 / eps must stay below ~2.6e5,
        i.e. eps >= 2e-5 extent^2 with a factor 3 of headroom.  The extent is always MEASURED on the data: a ``diameter=`` given by
        the caller only parametrises the schedule (``_legacy/sinkhorn_divergence.py:154-163``) and may understate the clouds, so
        it is never used here (round 5 trusted it with a factor 15 and returned inf / nan on understated values).
        p = 2 only; GEOMLOSS_HIP_F16X2=0 keeps bf16 x 3.

        (Tried in round 6 and dropped: explicit differences, GLHIP_FLAG_DIRECT, for the temperatures below extent^2 / 4e4.  At
        extent^2 / eps = 3.3e5 — blur / extent = 0.0017, tests/test_samples_loss_gpu.py::test_understated_diameter_is_legal — the
        gradient went from 6.2e-4 to 3.8e-4 of its max-norm only: in that near-assignment regime the float32 dual potentials of
        the earlier temperatures pass through unattenuated, whatever the last exponents are made of.  The loss is within 1e-4.)"""
        if self.p == 2 and _F16X2 and extent is not None and extent > 0:
            self.h2_min_eps = 2e-5 * float(extent) ** 2

    def _flags(self, eps):
        return hip.FLAG_F16X2 if eps >= self.h2_min_eps else 0

    def __call__(self, eps, C, h):
        if self.multiscale:
            return softmin_multiscale(eps, C, h, p=self.p, flags=self._flags(eps))
        return softmin_online(eps, C, h, p=self.p, flags=self._flags(eps))

    def step(self, eps, C, log_w, pot, damping, prev):
        x, y = C[0], C[1]
        ranges = C[4] if self.multiscale else None
        if not hip.fused_step_applies(x.shape[-1], self.p, 0, ranges is not None):  # no fused kernel on the generic-dimension path (incl. D > 3 under NO_MFMA / DIRECT)
            ft = damping * self(eps, C, log_w if pot is None else log_w + pot / eps)
            return ft if prev is None else 0.5 * (prev + ft)
        flat = (lambda t: None if t is None else t.reshape(-1)) if x.dim() == 2 else (lambda t: t)
        out = hip.sinkhorn_step(eps, x, y, flat(log_w), flat(pot), flat(prev), damping, p=self.p, ranges=ranges, flags=self._flags(eps))
        return out.view(1, -1) if (x.dim() == 2 and not self.multiscale) else out

    def value_and_grad(self, eps, C, log_w, pot_new, pot_old, f_new, f_old, damping):
        """The last, differentiable update ``damping * softmin(eps, C, log_w + pot_new / eps)`` with its gradient from ONE reduction
        (``glhip_softmin_fwd_grad``), or None.  The previous iteration of the loop ran the same soft-min on ``pot_old``: that value,
        ``(2 f_new - f_old) / damping``, is within ``sup |pot_new - pot_old|`` of the one wanted now (1-Lipschitz)."""
        x, y = C[0], C[1]
        if self.p != 2 or pot_old is None or f_old is None or not (torch.is_grad_enabled() and x.requires_grad):
            return None
        ranges = C[4] if self.multiscale else None
        flat = (lambda t: t.reshape(-1)) if x.dim() == 2 else (lambda t: t)
        guess = (2.0 * f_new - f_old) / damping
        margin = (pot_new - pot_old).abs().max()
        out = hip.softmin_value_and_grad(eps, x, y, flat(log_w + pot_new / eps), flat(guess), margin, ranges=ranges, flags=self._flags(eps))
        if out is None:
            return None
        out = damping * out
        return out.view(1, -1) if (x.dim() == 2 and not self.multiscale) else out

    def _iter4_plan(self, C_xy, a_log, b_log, debias, create):
        """The hip.Iter4Plan of the loop being run, (re)built when the inputs change; None when the one-launch-per-iteration
        path does not apply: block-sparse levels, D > 16, and problems big enough for every soft-min to fill the GPU
        on its own (those run faster as separate launches with pre-packed columns).  The coarse level of the multiscale
        backend (dense, ~2e3 clusters with their own weights) qualifies: its 7 x 4 soft-mins become 7 launches."""
        x, y = C_xy[0], C_xy[1]
        if self.p not in (1, 2) or x.shape[-1] > hip.XD_MAX_DIM or not _fuse_iterations or x.dtype == torch.float64:
            return None
        if self.p == 1 and (hip.autosort_applies(x, y) or hip.autosort_applies(y, x)):
            return None     # big p = 1 clouds: one voxel-sorted launch per soft-min (glhip_dist_x32.h) beats the fused iteration
        if hip.ENV_FLAGS & (hip.FLAG_NO_MFMA | hip.FLAG_DIRECT | hip.FLAG_F32_MFMA | hip.FLAG_XDL16):
            return None     # the one-launch iteration exists on the default kernel only: a kernel-selection flag means "not that one"
        if self.multiscale and C_xy[4] is not None:     # truncated fine level: block-sparse launches
            return None
        B = 1 if x.dim() == 2 else x.shape[0]
        if float(B) * x.shape[-2] * y.shape[-2] >= _ITER4_MAX_PAIRS:
            return None
        plan = self._plan
        if plan is None or plan[0] is not x or plan[1] is not a_log or plan[2] is not b_log or plan[3] != debias:
            if not create:
                return None
            plan = self._plan = (x, a_log, b_log, debias, hip.Iter4Plan(x, y, a_log, b_log, debias, p=self.p))
        return plan[4]

    def iter4(self, eps, C_xy, a_log, b_log, pots, damping, debias):
        """All simultaneous updates of one iteration in one launch (``glhip_sinkhorn_iter4``), or None (see _iter4_plan)."""
        plan = self._iter4_plan(C_xy, a_log, b_log, debias, create=True)
        if plan is None:
            return None
        plan.extra_flags = self._flags(eps)
        return plan.run(eps, damping, pots)

    def anneal(self, eps_list, dampings, C_xy, a_log, b_log, debias):
        """The initialisation and all the iterations of a level queued by one library call (``glhip_sinkhorn_anneal``), or None where
        :meth:`iter4` does not apply (see _iter4_plan) or `_anneal_in_library` is off.  Returns (final potentials, inputs of the last
        iteration).  Same launches as one :meth:`iter4` call per temperature — the f16 x 2 layout from the same temperature on —
        without the Python interpreter between them: what bounds a loop on a few thousand points."""
        if not _anneal_in_library:
            return None
        plan = self._iter4_plan(C_xy, a_log, b_log, debias, create=True)
        if plan is None:
            return None
        return plan.anneal(eps_list, dampings, self.h2_min_eps)

    def extrapolate4(self, pots, eps, damping, C_xy, C_yx, a_log, b_log, C_xy_fine, C_yx_fine, debias):
        """The coarse-to-fine jump of every potential as one launch (``glhip_sinkhorn_extrapolate4``), or None where the fused
        iterations do not apply either (D > 16, float64 clouds, a kernel-selection flag in the environment, fusion switched off).
        ``a_log`` / ``b_log``: the log-weights of the COARSE measures, ``pots`` the potentials on them."""
        x_, y_, xc, yc = C_xy_fine[0], C_yx_fine[0], C_xy[0], C_yx[0]
        if (self.p not in (1, 2) or x_.shape[-1] > hip.XD_MAX_DIM or not _fuse_iterations
                or any(t.dtype == torch.float64 for t in (x_, y_, xc, yc))
                or hip.ENV_FLAGS & (hip.FLAG_NO_MFMA | hip.FLAG_DIRECT | hip.FLAG_F32_MFMA | hip.FLAG_XDL16)):
            return None
        return hip.sinkhorn_extrapolate4(eps, x_, y_, xc, yc, a_log, b_log, pots, damping, flags=self._flags(eps), p=self.p)

    def last4(self, eps, C_xy, C_yx, a_log, b_log, pots, damping, debias, create=False):
        """The differentiable, non-averaged last update of every potential as one forward launch (and one autograd node);
        None when :meth:`iter4` did not run this loop (unless ``create``)."""
        plan = self._iter4_plan(C_xy, a_log, b_log, debias, create=create)
        if plan is None:
            return None
        plan.extra_flags = self._flags(eps)
        return hip.sinkhorn_last4(plan, C_xy[0], C_yx[0], eps, damping, pots)


# hipGraph mode for the launch-bound regime (small clouds): the whole autograd-free annealing loop of the online
# backend — 4 fused launches per temperature — is captured once per (shapes, schedule) and replayed as ONE graph launch.
# Opt-in (GEOMLOSS_HIP_GRAPH=1 or set_graph_mode(True)) and only when `diameter` is given, because the temperatures are
# kernel arguments baked into the graph: a data-dependent diameter would force a new capture for every input.
_graph_mode = os.environ.get("GEOMLOSS_HIP_GRAPH", "0") == "1"
_COARSE_F64_MIN_PAIRS = float(os.environ.get("GEOMLOSS_HIP_COARSE_F64_MIN_PAIRS", "1e11"))   # two-scale losses: float64 coarse level from here on
_F16X2 = os.environ.get("GEOMLOSS_HIP_F16X2", "1") != "0"      # f16 x 2 exponents where the temperature allows (_HipSoftmin.set_range)
_anneal_in_library = True   # the iterations of a level queued by one library call (tests switch it off to compare with one call per iteration)
_fuse_iterations = True   # one launch per Sinkhorn iteration (small / mid-size clouds); set_iteration_fusion(False): four half-steps
# ... up to this many pairs per soft-min; bigger problems fill the GPU with one soft-min per launch (pre-packed columns, XCD grids)
_ITER4_MAX_PAIRS = 4e9   # measured: B x 4096^2 with B = 32..128 and N = 3e4 gain 5-13 %, 7e4+ lose
_graphs = hip.GraphCache()


def set_graph_mode(enabled):
    global _graph_mode
    _graph_mode = bool(enabled)


def graph_mode():
    return _graph_mode


def set_iteration_fusion(enabled):
    """One ``glhip_sinkhorn_iter4`` launch per iteration of the online loop (default) or four ``glhip_sinkhorn_step``."""
    global _fuse_iterations
    _fuse_iterations = bool(enabled)


# A given `diameter` says nothing reliable about the data (set_range): online losses of at least this many pairs per soft-min measure
# the bounding box themselves — one small reduction and its read-back, < 1 % of such a loss — smaller ones stay on bf16 x 3, where
# the host bounds the run time anyway and a synchronisation would cost more than the layout gains.
_EXTENT_MIN_PAIRS = 2e8


def _extent_for_range(x, y, diameter, diameter_given, bounds_the_data=False):
    """The extent `_HipSoftmin.set_range` may rely on: the measured diameter; a given one only if the caller vouches that it bounds
    the data (``ShardedSamplesLoss``: the all-reduced bounding box); otherwise a measurement for big problems, None for small ones."""
    if not diameter_given or bounds_the_data:
        return diameter
    B = 1 if x.dim() == 2 else x.shape[0]
    if not _F16X2 or _graph_mode or float(B) * x.shape[-2] * y.shape[-2] < _EXTENT_MIN_PAIRS:
        return None
    D = x.shape[-1]
    if x.dtype in (torch.bfloat16, torch.float16):
        x, y = x.float(), y.float()
    return max_diameter(x.detach().reshape(-1, D), y.detach().reshape(-1, D))


def sinkhorn_online(
    a, x, b, y, p=2, blur=0.05, reach=None, diameter=None, scaling=0.5, cost=None, debias=True,
    potentials=False, **kwargs,
):
    """Sinkhorn divergence with O(N+M) memory; a (B,N), x (B,N,D), b (B,M), y (B,M,D) on a GPU."""
    B = x.shape[0]
    diameter_given = diameter is not None
    a, b = _fp32_weights(a, b)
    if cost is not None:
        if B > 1:
            raise ValueError("Custom cost functions are not yet supported with batches." "")
        p = _exponent_of(cost)
    if B == 1:  # like the reference, the single-problem path works on (N,D) clouds
        x, y = x.squeeze(0), y.squeeze(0)
    softmin = _HipSoftmin(p, multiscale=False)

    C_xx, C_yy = ((x, x.detach()), (y, y.detach())) if debias else (None, None)
    C_xy, C_yx = ((x, y.detach()), (y, x.detach()))

    diameter, eps, eps_list, rho = scaling_parameters(x, y, p, blur, reach, diameter, scaling)
    softmin.set_range(_extent_for_range(x, y, diameter, diameter_given, kwargs.get("diameter_bounds_the_data", False)))

    a_log, b_log = log_weights_many([a, b])
    # (p = 1 on clouds big enough for the voxel-sorted distance plans: those are built with a host read-back, which a stream
    # capture does not allow — and launches of that size gain nothing from a graph)
    sorts = p == 1 and (hip.autosort_applies(x, y) or hip.autosort_applies(y, x))
    if _graph_mode and diameter_given and x.is_cuda and x.shape[-1] <= 3 and not sorts and x.dtype != torch.float64:
        f_aa, g_bb, g_ab, f_ba = _graphed_loop(softmin, x, y, a_log, b_log, eps_list, rho, debias)
    else:
        f_aa, g_bb, g_ab, f_ba = sinkhorn_loop(
            softmin, a_log, b_log, C_xx, C_yy, C_xy, C_yx, eps_list, rho, debias=debias
        )
    return sinkhorn_cost(eps, rho, a, b, f_aa, g_bb, g_ab, f_ba, batch=True, debias=debias, potentials=potentials)


def _graphed_loop(softmin, x, y, a_log, b_log, eps_list, rho, debias):
    """`sinkhorn_loop` with its autograd-free part replayed from a hipGraph; the last, differentiable half-steps
    (sinkhorn_divergence.py:612-623) run eagerly, exactly as in the loop."""
    from .sinkhorn_divergence import dampening

    def annealing(xs, ys, al, bl):
        # A fresh soft-min object per invocation: its Iter4Plan (fp32 / contiguous copies of the clouds, scratch, ping-pong
        # potentials) is then built INSIDE the captured region, so the copies are re-made from the static inputs on every
        # replay and every buffer lives in the graph's private pool (a plan left over from the warm-up pass would freeze the
        # first call's converted clouds into the graph and leave the replays writing into freed memory).
        sm = _HipSoftmin(softmin.p, multiscale=False)
        sm.h2_min_eps = softmin.h2_min_eps
        Cxx, Cyy = ((xs, xs), (ys, ys)) if debias else (None, None)
        out = sinkhorn_loop(sm, al, bl, Cxx, Cyy, (xs, ys), (ys, xs), eps_list, rho, debias=debias,
                            last_extrapolation=False)
        return out

    key = (tuple(x.shape), tuple(y.shape), tuple(x.stride()), tuple(y.stride()), x.dtype, y.dtype, x.device.index,
           tuple(float(e) for e in eps_list), rho, softmin.p, debias)
    f_aa, g_bb, g_ab, f_ba = _graphs.run(key, annealing, (x, y, a_log, b_log))
    torch.autograd.set_grad_enabled(True)   # what sinkhorn_loop leaves behind (reference behaviour)
    eps = eps_list[-1]
    damping = dampening(eps, rho)
    xd, yd = x.detach(), y.detach()
    fused = softmin.last4(eps, (x, yd), (y, xd), a_log, b_log, (f_ba, g_ab, f_aa, g_bb) if debias else (f_ba, g_ab), damping,
                          debias, create=True)
    if fused is not None:   # same kernels as the eager loop's last step
        f_ba, g_ab = fused[0], fused[1]
        if debias:
            f_aa, g_bb = fused[2], fused[3]
    else:
        f_ba, g_ab = (damping * softmin(eps, (x, yd), (b_log + g_ab / eps).detach()),
                      damping * softmin(eps, (y, xd), (a_log + f_ba / eps).detach()))
        if debias:
            f_aa = damping * softmin(eps, (x, xd), (a_log + f_aa / eps).detach())
            g_bb = damping * softmin(eps, (y, yd), (b_log + g_bb / eps).detach())
    return f_aa, g_bb, g_ab, f_ba


# ==============================================================================
#                          backend == "multiscale"
# ==============================================================================


def softmin_multiscale(eps, C_xy, f_y, p=2, flags=0):
    """Block-sparse soft-C-transform (``:445-450``): C_xy = (x, y, ranges_x, ranges_y, ranges_xy)."""
    x, y, ranges_x, ranges_y, ranges_xy = C_xy
    return hip.softmin(eps, x, y, f_y.view(-1), p=p, ranges=ranges_xy, flags=flags)


def clusterize(a, x, scale=None, labels=None):
    """Voxel-grid clustering of a weighted cloud (``:453-490``).

    Returns ``[a_c, a], [x_c, x], [ranges_x], perm``: cluster weights / centroids, the cloud re-ordered
    so that cluster k is ``x[ranges_x[k,0]:ranges_x[k,1]]``, and the permutation that was applied.
    """
    if labels is None and scale is None:
        return [a], [x], []
    if native_clustering_applies(x, labels):     # one C-ABI call instead of ~25 torch launches and two host round trips
        a_c, a_s, x_c, x_s, ranges_x, perm = clusterize_device(a, x, scale)
        return [a_c, a_s], [x_c, x_s], [ranges_x], perm
    x_lab = grid_cluster(x, scale) if labels is None else labels
    _, perm = torch.sort(x_lab.view(-1), stable=True)
    ranges_x, x_c, a_c = cluster_ranges_centroids(x, x_lab, weights=a, perm=perm)
    return [a_c, a[perm]], [x_c, x[perm]], [ranges_x], perm


# A truncated fine level is an optimisation, and not always one.  The reference's voxel rule makes ~2000 clusters whatever N is, so
# below ~3e4 points a cluster holds a handful of them and a block-sparse launch fills a fraction of its 32-row tiles (N = 1e4, D = 3:
# 99 us per soft-min against 10 us for the dense kernel on the same points); and labels that leave a coordinate out (the reference's
# 4-D recipe clusters the 3 spatial coordinates of position + feature points) give clusters as wide as the cloud, of which the rule
# keeps 86 %.  Before the pattern is built the pairs its rule keeps are counted (one launch, one 24-byte read-back) and costed against
# the dense launch:
#     block-sparse ~ kept / (kDenseRate * fill),   fill = filled share of the 32-row tiles of a mean cluster,   dense ~ N M
# and the cheaper one runs — the fine level of `truncate=None` (``:504-505``).  Only where that cannot move a result: a dropped
# pair has an exponent below -truncate at the temperature of the jump and, the loop annealing on, below -truncate * eps_jump / eps_last
# in the iterations that decide the answer; the switch needs that bound to be 16 (1e-7 of a row's mass: float32 resolution).  The
# defaults qualify for p = 2 (5 x (0.108 / 0.05)^2 = 23 on the unit cube), not for p = 1 (5 x 2.2 = 11), which keeps its pattern.
# GEOMLOSS_HIP_DENSE_SWITCH=0 keeps the pattern everywhere (A/B, tests).
_DENSE_SWITCH = os.environ.get("GEOMLOSS_HIP_DENSE_SWITCH", "1")
_DENSE_SWITCH_MIN_EXPONENT = 16.0
_DENSE_SWITCH_RATE = 0.45         # pairs per second of a block-sparse launch with full tiles, as a share of the dense kernel's (fitted: profiles/r04_dense_switch.txt)


def set_dense_switch(mode):
    """"1": cost model, "0": always block-sparse, "always": always dense (tests)."""
    global _DENSE_SWITCH
    _DENSE_SWITCH = str(mode)


def dense_is_cheaper(kept, N, M, Cr, Cc):
    """The cost model above for both orientations of a pattern (N x M points, Cr x Cc clusters, `kept` pairs of points)."""
    if N <= 0 or M <= 0 or Cr <= 0 or Cc <= 0:
        return False

    def fill(points, clusters):
        mean = points / clusters
        return mean / (32.0 * math.ceil(mean / 32.0))

    sparse = 0.5 * kept / _DENSE_SWITCH_RATE * (1.0 / fill(N, Cr) + 1.0 / fill(M, Cc))
    return sparse > float(N) * float(M)


# rows of a 2-wavefront workgroup of the block-sparse soft-min (GLHIP_FLAG_SMALL_ROW_BLOCKS); 0: never hint
_SMALL_ROW_BLOCK = 64


def _goes_dense(truncate, eps, eps_last, N, M, Cr, Cc, kept_pairs):
    """``(dense, small_rows, small_cols)``.  `kept_pairs`: callable -> (kept pairs, sum of squared row-cluster sizes, of column-cluster
    sizes); the read-back is only paid where the switch may apply.  small_*: the block a typical PAIR lives in (sum of squares /
    points — not the mean cluster: a cloud sampled on a surface has clusters of 47 points on average at N = 1e5 and most of its pairs
    in clusters of hundreds) has at most 64 points, the launch hint of hip.BlockRanges."""
    if _DENSE_SWITCH == "always":
        return True, False, False
    if _DENSE_SWITCH == "0" or eps_last is None or truncate * eps / eps_last < _DENSE_SWITCH_MIN_EXPONENT:
        return False, False, False
    kept, sq_rows, sq_cols = kept_pairs()
    return dense_is_cheaper(kept, N, M, Cr, Cc), sq_rows <= _SMALL_ROW_BLOCK * N, sq_cols <= _SMALL_ROW_BLOCK * M


def kernel_truncation_prefetch(calls, eps, truncate=None, cost=None, eps_last=None):
    """The keep-rule counts that the :func:`kernel_truncation` calls of one coarse-to-fine jump will ask for — the cross term and the
    two debiasing terms — queued together and read back in ONE host round trip.  ``calls``: [(C, C_t, f, g), ...] as they will be
    passed to kernel_truncation; returns a list of ``kept`` triples (None where the call would not count)."""
    if truncate is None or _DENSE_SWITCH in ("0", "always") or eps_last is None or truncate * eps / eps_last < _DENSE_SWITCH_MIN_EXPONENT:
        return [None] * len(calls)
    native_p = getattr(cost, "glhip_exponent", None)
    pending = []
    for C, C_t, f, g in calls:
        x, y = C[0], C_t[0]
        ok = native_p is not None and native_keep_rule_applies(x)
        pending.append(kept_pairs_device("dual_slack", x, y, f, g, C[2], C[3], truncate * eps, p=native_p, defer=True) if ok else None)
    live = [t for t in pending if t is not None]
    values = iter(hip.read_back(*live)) if live else iter(())
    return [None if t is None else tuple(next(values)) for t in pending]


def kernel_truncation(C_xy, C_yx, C_xy_, C_yx_, f_ba, g_ab, eps, truncate=None, cost=None, verbose=False, eps_last=None, kept=None):
    """Keeps the fine blocks whose coarse dual slack allows mass: f_i + g_j > C_ij - truncate * eps (``:493-530``).
    ``eps_last`` (not in the reference): the last temperature of the loop, which lets the fine level stay dense where that is cheaper
    and changes nothing (see above); None: always the pattern.  ``kept``: the counts of :func:`kernel_truncation_prefetch`."""
    if truncate is None:
        return C_xy_, C_yx_
    x, yd, ranges_x, ranges_y, _ = C_xy
    y, xd, _, _, _ = C_yx
    x_, yd_, ranges_x_, ranges_y_, _ = C_xy_
    y_, xd_, _, _, _ = C_yx_
    native_p = getattr(cost, "glhip_exponent", None)     # the two built-in costs carry their exponent
    if native_p is not None and native_keep_rule_applies(x):
        rule = ("dual_slack", x, y, f_ba, g_ab, ranges_x, ranges_y, truncate * eps)
        dense, small_x, small_y = _goes_dense(truncate, eps, eps_last, x_.shape[0], y_.shape[0], x.shape[0], y.shape[0],
                                              (lambda: kept) if kept is not None else (lambda: kept_pairs_device(*rule, p=native_p)))
        # the debiasing terms hand in one cloud and one potential twice (sinkhorn_divergence.py:284-289): a symmetric pattern
        symmetric = C_xy is C_yx and f_ba is g_ab
        ranges_xy_ = None if dense else block_ranges_device(*rule, p=native_p, symmetric=symmetric)
        if ranges_xy_ is not None:
            ranges_xy_.small_i, ranges_xy_.small_j = small_x, small_y
        if verbose:     # the printed statistic only: the ranges above are the ones a silent run builds (same kernels either way)
            with torch.no_grad():
                C = cost(x, y)
                ks, Cs = (f_ba.view(-1, 1) + g_ab.view(1, -1) > C - truncate * eps).sum(), C.shape[0] * C.shape[1]
            print("Keep {}/{} = {:2.1f}% of the coarse cost matrix.".format(ks, Cs, 100 * float(ks) / Cs))
        if dense:      # dense fine level: the cost objects of a single-scale loop
            return (x_, yd_, None, None, None), (y_, xd_, None, None, None)
        return (x_, yd_, ranges_x_, ranges_y_, ranges_xy_), (y_, xd_, ranges_y_, ranges_x_, swap_axes(ranges_xy_))
    with torch.no_grad():
        C = cost(x, y)
        keep = f_ba.view(-1, 1) + g_ab.view(1, -1) > C - truncate * eps
        if verbose:
            ks, Cs = keep.sum(), C.shape[0] * C.shape[1]
            print("Keep {}/{} = {:2.1f}% of the coarse cost matrix.".format(ks, Cs, 100 * float(ks) / Cs))

        def kept_pairs():
            rows = (ranges_x[:, 1] - ranges_x[:, 0]).double()
            cols = (ranges_y[:, 1] - ranges_y[:, 0]).double()
            return tuple(torch.stack([rows @ (keep.double() @ cols), rows @ rows, cols @ cols]).tolist())

        dense, small_x, small_y = (_goes_dense(truncate, eps, eps_last, x_.shape[0], y_.shape[0], x.shape[0], y.shape[0], kept_pairs)
                                   if x_.is_cuda else (False, False, False))
        if dense:
            return (x_, yd_, None, None, None), (y_, xd_, None, None, None)
        ranges_xy_ = from_matrix(ranges_x, ranges_y, keep)
        ranges_xy_.small_i, ranges_xy_.small_j = small_x, small_y
    return (x_, yd_, ranges_x_, ranges_y_, ranges_xy_), (y_, xd_, ranges_y_, ranges_x_, swap_axes(ranges_xy_))


def _truncation(verbose, eps_last):
    """kernel_truncation as the loop calls it, plus its `prefetch` hook (sinkhorn_divergence.sinkhorn_loop: one read-back per jump)."""
    fn = partial(kernel_truncation, verbose=verbose, eps_last=eps_last)
    fn.prefetch = partial(kernel_truncation_prefetch, eps_last=eps_last)
    return fn


def extrapolate_samples(f_ba, g_ab, eps, damping, C_xy, b_log, C_xy_, softmin=None):
    """Coarse-to-fine update of a potential: one soft-min of the fine points against the coarse measure (``:533-544``)."""
    yd = C_xy[1]  # coarse source points
    x_ = C_xy_[0]  # fine target points
    return damping * softmin(eps, (x_, yd, None, None, None), (b_log + g_ab / eps).detach())


def sinkhorn_multiscale(
    a, x, b, y, p=2, blur=0.05, reach=None, diameter=None, scaling=0.5, truncate=5, cost=None,
    cluster_scale=None, debias=True, potentials=False, labels_x=None, labels_y=None, verbose=False, **kwargs,
):
    """Two-scale Sinkhorn divergence; a (N,), x (N,D), b (M,), y (M,D) on a GPU (``:547-681``)."""
    N, D = x.shape
    a, b = _fp32_weights(a, b)
    if cost is None:
        cost = cost_formulas[p], cost_routines[p]
    cost_formula, cost_routine = cost[0], cost[1]
    p_kernel = _exponent_of(cost_formula)
    softmin = _HipSoftmin(p_kernel, multiscale=True)
    extrapolate = partial(extrapolate_samples, softmin=softmin)
    extrapolate.all4 = softmin.extrapolate4     # the loop's one-launch hook (sinkhorn_divergence.sinkhorn_loop)

    diameter_given = diameter is not None
    diameter, eps, eps_list, rho = scaling_parameters(x, y, p, blur, reach, diameter, scaling)

    # voxel size: about 2000 cells over the bounding box
    if cluster_scale is None:
        cluster_scale = diameter / (np.sqrt(D) * 2000 ** (1 / D))
    if native_clustering_applies(x, labels_x) and native_clustering_applies(y, labels_y):      # both clusterings, one host round trip
        box = []         # ... which also brings back the voxel bounds of the clouds: their true extent when `diameter` was given
        (a_c, a, x_c, x, ranges_x, perm_x), (b_c, b, y_c, y, ranges_y, perm_y) = clusterize_device_many(
            [(a, x), (b, y)], cluster_scale, long_perm=False, extent=box)
        softmin.set_range(box[0] if diameter_given else diameter)
    else:
        softmin.set_range(None if diameter_given else diameter)
        [a_c, a], [x_c, x], [ranges_x], perm_x = clusterize(a, x, scale=cluster_scale, labels=labels_x)
        [b_c, b], [y_c, y], [ranges_y], perm_y = clusterize(b, y, scale=cluster_scale, labels=labels_y)

    # Switch to the fine clouds once the blur radius drops below the voxel size.
    # N.B.: like the reference (``:593-597``) the search variable is named `eps`, so the temperature
    # handed to sinkhorn_cost below is the one at which the search stopped.  The balanced formulas do
    # not use it.
    jumps = [len(eps_list) - 1]
    for i, eps in enumerate(eps_list[2:]):
        if cluster_scale**p > eps:
            jumps = [i + 1]
            break

    if verbose:
        print("{}x{} clusters, computed at scale = {:2.3f}".format(len(x_c), len(y_c), cluster_scale))
        print("Successive scales : ", ", ".join(["{:.3f}".format(x ** (1 / p)) for x in eps_list]))
        if jumps[0] >= len(eps_list) - 1:
            print("Extrapolate from coarse to fine after the last iteration.")
        else:
            print(
                "Jump from coarse to fine between indices {} (σ={:2.3f}) and {} (σ={:2.3f}).".format(
                    jumps[0], eps_list[jumps[0]] ** (1 / p), jumps[0] + 1, eps_list[jumps[0] + 1] ** (1 / p)
                )
            )

    if float(N) * y.shape[0] >= _COARSE_F64_MIN_PAIRS and x_c.dtype == torch.float32:
        # The coarse level of a big problem runs in float64.  Two samples of one law at N = 1e6 leave a gradient of 1e-3 of a blur and a
        # loss of 1e-4 of its terms, and the float32 rounding of the coarse trajectory (potentials of size 1 at the first temperatures:
        # 1e-7 per cluster and iteration) reaches them through the nearly undamped mode (f + c(x), g - c(y)): dL/dx was 4e-5 .. 6e-5 of
        # its max-norm away from a float64 run (profiles/r06_full_size_parity.txt).  ~2000 clusters: 36 float64 soft-mins of 35 us
        # (glhip_api_f64.hip spreads their rows over wavefronts) plus their torch arithmetic, < 1 % of such a loss.
        a_c, b_c, x_c, y_c = a_c.double(), b_c.double(), x_c.double(), y_c.double()
    la_c, la, lb_c, lb = log_weights_many([a_c, a, b_c, b])
    a_logs, b_logs = [la_c, la], [lb_c, lb]
    if debias:
        C_xxs = [(x_c, x_c.detach(), ranges_x, ranges_x, None), (x, x.detach(), None, None, None)]
        C_yys = [(y_c, y_c.detach(), ranges_y, ranges_y, None), (y, y.detach(), None, None, None)]
    else:
        C_xxs = C_yys = None
    C_xys = [(x_c, y_c.detach(), ranges_x, ranges_y, None), (x, y.detach(), None, None, None)]
    C_yxs = [(y_c, x_c.detach(), ranges_y, ranges_x, None), (y, x.detach(), None, None, None)]

    f_aa, g_bb, g_ab, f_ba = sinkhorn_loop(
        softmin, a_logs, b_logs, C_xxs, C_yys, C_xys, C_yxs, eps_list, rho,
        jumps=jumps, cost=cost_routine, kernel_truncation=_truncation(verbose, eps_list[-1]),
        truncate=truncate, extrapolate=extrapolate, debias=debias,
    )

    cost = sinkhorn_cost(eps, rho, a, b, f_aa, g_bb, g_ab, f_ba, debias=debias, potentials=potentials)

    if potentials:  # undo the cluster sort
        F_x, G_y = cost
        f_x, g_y = F_x.clone(), G_y.clone()
        f_x[perm_x.long()], g_y[perm_y.long()] = F_x, G_y
        return f_x, g_y
    return cost
