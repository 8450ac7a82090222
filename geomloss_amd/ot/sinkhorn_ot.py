"""Annealing schedule, symmetric Sinkhorn loop and loss formulas of the ``geomloss.ot`` solvers, on the HIP kernels.

Host-side restatement (stays in Python, as BASELINE.json prescribes for the outer loop) of
``ot/_abstract_solvers/annealing.py:18-225`` (``max_diameter``, ``annealing_parameters``),
``ot/_abstract_solvers/sinkhorn_ot.py:17-30,32-447`` (``sinkhorn_initialization``, ``sinkhorn_loop``; single scale) and
``ot/_abstract_solvers/unbalanced_ot.py:14-192`` (``dampening``, ``sinkhorn_cost``) of the reference.

The soft-min is *not* passed in as a callable here: the loop talks to the kernels through a small cost object
(:class:`SampleCost`) so that one iteration is ONE launch (``glhip_sinkhorn_iter4``) where the fused kernel applies and
four fused half-steps (``glhip_sinkhorn_step``) otherwise.

Units.  The solvers of ``geomloss.ot`` use the cost C = |x - y|^2 *without* the 1/2 of the legacy API, and a temperature
``reg``.  The kernels implement C/2; since  softmin_eps(C, g) = 2 softmin_{eps/2}(C/2, g/2)  the whole loop runs in
"half units" (eps/2, rho/2, potentials/2) and :func:`sinkhorn_loop` doubles the potentials it returns.  No coordinate
is rescaled.
"""

from typing import List, NamedTuple, Optional

import numpy as np
import torch

from .. import hip


class DescentParameters(NamedTuple):
    scale_list: List[int]
    eps_list: List[float]
    rho_list: List[Optional[float]]


class SinkhornPotentials(NamedTuple):
    g_ab: torch.Tensor
    f_ba: torch.Tensor
    f_aa: Optional[torch.Tensor] = None
    g_bb: Optional[torch.Tensor] = None


def max_diameter(x, y):
    """Diagonal of the joint bounding box of x (N,D) and y (M,D) (``annealing.py:18-35``)."""
    mins = torch.minimum(x.amin(0), y.amin(0))
    maxs = torch.maximum(x.amax(0), y.amax(0))
    return float((maxs - mins).double().norm())


def annealing_parameters(*, maxmin_cost, eps, rho=None, n_iter=None, scaling=None, eps_scales=None):
    """Temperatures of the loop (``annealing.py:47-225``): geometric progression from ``maxmin_cost`` to ``eps``."""
    if n_iter is not None and n_iter <= 0:
        raise ValueError("The number of iterations should be >= 1. " f"Received n_iter={n_iter}.")
    if scaling is not None and (scaling <= 0 or scaling > 1):
        raise ValueError("The scaling factor should be in (0,1]. " f"Received scaling={scaling}.")
    if n_iter is None and scaling is None:
        raise ValueError("Please specify a number of iterations using either " "the n_iter or scaling parameters.")

    maxmin_cost = max(float(maxmin_cost), eps)
    if n_iter is None:
        if scaling == 1:
            raise ValueError(
                "If n_iter is not specified, the scaling coefficient "
                "should be < 1. Keeping a constant value for the temperature epsilon "
                "(with scaling = 1) does not allow us to stop convergence and may lead "
                "to an infinite loop."
            )
        n_iter = int(np.floor((np.log(eps) - np.log(maxmin_cost)) / np.log(scaling))) + 2

    if scaling == 1:
        eps_list = [eps] * n_iter
    elif scaling is None:
        eps_list = [eps] if n_iter == 1 else np.geomspace(maxmin_cost, eps, n_iter)
    else:
        eps_list = np.exp(np.maximum(np.log(maxmin_cost) + np.arange(n_iter) * np.log(scaling), np.log(eps)))
    eps_list = [float(e) for e in eps_list]
    rho_list = [rho] * len(eps_list)

    if eps_scales is None or len(eps_scales) < 2:
        scale_list = [0] * len(eps_list)
    else:
        scale_list, scale = [], 0
        for e in eps_list:
            while scale + 1 < len(eps_scales) and e < eps_scales[scale]:
                scale += 1
            scale_list.append(scale)
        scale_list[-1] = len(eps_scales) - 1
    return DescentParameters(scale_list=scale_list, eps_list=eps_list, rho_list=rho_list)


def damping_factor(eps, rho):
    """``dampening`` (``unbalanced_ot.py:14-19``) as a multiplier: 1 / (1 + eps/rho), 1 for balanced OT."""
    return 1.0 if rho is None else 1.0 / (1.0 + eps / rho)


class SampleCost:
    """The four implicit cost matrices C(x_i, y_j) = |x_i - y_j|^2 of one problem, as the kernels see them: two clouds.
    Stands in for the ``CostMatrices`` of LazyTensors built at ``ot/_implementations/sample.py:367-378``."""

    def __init__(self, x, y):
        self.x, self.y = x, y
        self.xd, self.yd = x.detach(), y.detach()

    def pair(self, which, grad):
        """(rows, columns) of cost ``which`` in {"xy","yx","xx","yy"}; the column cloud never carries gradients."""
        rows = {"x": self.x if grad else self.xd, "y": self.y if grad else self.yd}[which[0]]
        cols = {"x": self.xd, "y": self.yd}[which[1]]
        return rows, cols


def _mean_cost(rows, cols, w):
    """sum_j w_j |r_i - c_j|^2 / (2 sum_j w_j)  — the eps = +inf soft-min of ``softmin_sample`` (``sample.py:139-154``) for a
    zero potential, in half units.  The sum over j factorises into three moments of the weighted column cloud, so this is an
    O(N + M) computation (float64 torch on the GPU, clouds centred): no N x M reduction is needed."""
    r, c, w = rows.detach().double(), cols.detach().double(), w.detach().double()
    centre = c.mean(0, keepdim=True)
    r, c = r - centre, c - centre
    sw = w.sum()
    m1 = (w[:, None] * c).sum(0)                      # sum_j w_j c_j
    m2 = (w * (c * c).sum(1)).sum()                   # sum_j w_j |c_j|^2
    return (0.5 * ((r * r).sum(1) * sw - 2.0 * (r @ m1) + m2) / sw).float()


def sinkhorn_loop(*, cost, log_a, log_b, descent, debias=True, last_extrapolation=True):
    """Symmetric Sinkhorn loop with annealing (``sinkhorn_ot.py:32-447``, single scale) -> :class:`SinkhornPotentials`.

    ``cost``: :class:`SampleCost`; ``log_a`` (N,), ``log_b`` (M,) fp32 on the GPU; ``descent``: temperatures / marginal
    strengths in the units of the ``geomloss.ot`` API (cost |x-y|^2).  Autograd: like the reference, the loop runs with
    gradients disabled and only the last, non-averaged update is recorded (differentiable in the row cloud of each cost).
    """
    x, y = cost.xd, cost.yd
    a, b = log_a.exp(), log_b.exp()
    prev_grad = torch.is_grad_enabled()
    torch.set_grad_enabled(False)
    try:
        eps, rho = descent.eps_list[0] / 2, None if descent.rho_list[0] is None else descent.rho_list[0] / 2
        lam = damping_factor(eps, rho)

        # sinkhorn_initialization (:17-30).  N.B.: un-batched, the reference's `bk.dot_products(exp(log_a), f)` reshapes its
        # 1-D arguments to (N, 1) and returns the ELEMENTWISE products a_i f_i, so the "constant offset" it subtracts is
        # 0.5 a_i f_i per point.  Reproduced as is: with few iterations the result depends on the starting point.
        def init(rows, cols, w_rows, w_cols):
            f = _mean_cost(rows, cols, w_cols)
            return lam * (f - 0.5 * w_rows * f)

        f_ba, g_ab = init(x, y, a, b), init(y, x, b, a)
        f_aa, g_bb = (init(x, x, a, a), init(y, y, b, b)) if debias else (None, None)

        plan, before = None, None
        fusable = (x.shape[1] <= hip.XD_MAX_DIM and float(x.shape[0]) * y.shape[0] < 4e9
                   and not (hip.ENV_FLAGS & (hip.FLAG_NO_MFMA | hip.FLAG_DIRECT | hip.FLAG_F32_MFMA | hip.FLAG_XDL16)))   # default kernel only
        if fusable:   # one launch per iteration (glhip_sinkhorn_iter4)
            plan = hip.Iter4Plan(x, y, log_a, log_b, debias)

        for eps, rho in zip(descent.eps_list, descent.rho_list):
            eps, rho = eps / 2, None if rho is None else rho / 2
            lam = damping_factor(eps, rho)
            pots = (f_ba, g_ab, f_aa, g_bb) if debias else (f_ba, g_ab)
            before = pots       # what the last iteration of the loop started from (see `last` below)
            if plan is not None:
                new = plan.run(eps, lam, tuple(p.view(1, -1) for p in pots))
                new = tuple(t.view(-1) for t in new)
            else:
                step = lambda r, c, lw, pot, prev: _averaged(eps, lam, r, c, lw, pot, prev)   # noqa: E731
                new = (step(x, y, log_b, g_ab, f_ba), step(y, x, log_a, f_ba, g_ab))
                if debias:
                    new += (step(x, x, log_a, f_aa, f_aa), step(y, y, log_b, g_bb, g_bb))
            f_ba, g_ab = new[0], new[1]
            if debias:
                f_aa, g_bb = new[2], new[3]
    finally:
        torch.set_grad_enabled(prev_grad)

    if last_extrapolation:   # :421-432 — coupled, non-averaged update on detached dual vectors
        def last(which, lw, pot, pot_old, f_new, f_old):
            rows, cols = cost.pair(which, grad=True)
            h = (lw + pot / eps).detach()
            # The last iteration of the loop ran this soft-min on pot_old: its value (2 f_new - f_old) / lam is within
            # sup |pot - pot_old| of the one wanted now, which lets big launches produce value and gradient in one reduction
            # (hip.softmin_value_and_grad; None when that does not apply).
            out = None if before is None else hip.softmin_value_and_grad(
                eps, rows, cols, h, (2.0 * f_new - f_old) / lam, (pot - pot_old).abs().max())
            return lam * (hip.softmin(eps, rows, cols, h) if out is None else out.view(-1))

        old = (None,) * 4 if before is None else before
        f_ba, g_ab = (last("xy", log_b, g_ab, old[1], f_ba, old[0]), last("yx", log_a, f_ba, old[0], g_ab, old[1]))
        if debias:
            f_aa, g_bb = (last("xx", log_a, f_aa, old[2], f_aa, old[2]), last("yy", log_b, g_bb, old[3], g_bb, old[3]))

    dbl = lambda t: None if t is None else 2.0 * t   # noqa: E731  (back to the units of C = |x-y|^2)
    return SinkhornPotentials(g_ab=dbl(g_ab), f_ba=dbl(f_ba), f_aa=dbl(f_aa), g_bb=dbl(g_bb))


def _averaged(eps, lam, rows, cols, log_w, pot, prev):
    """(prev + lam * softmin(eps, C, log_w + pot/eps)) / 2 as one fused launch (D <= 16 on the default kernels) or soft-min +
    torch arithmetic."""
    if hip.fused_step_applies(rows.shape[1], 2):
        return hip.sinkhorn_step(eps, rows, cols, log_w, pot, prev, lam)
    return 0.5 * (prev + lam * hip.softmin(eps, rows, cols, log_w + pot / eps))


def sinkhorn_cost(*, a, b, potentials, eps, rho, debias=True):
    """Value of the (debiased) Sinkhorn cost from the dual potentials, un-batched (``unbalanced_ot.py:22-176``)."""
    f_aa, g_bb, g_ab, f_ba = potentials.f_aa, potentials.g_bb, potentials.g_ab, potentials.f_ba
    assert f_ba.shape == a.shape and g_ab.shape == b.shape and eps > 0 and (rho is None or rho > 0)
    if rho is None:
        F_a, G_b = (f_ba - f_aa, g_ab - g_bb) if debias else (f_ba, g_ab)
    elif not debias:
        # Proposition 12 of Sejourne et al. 2019; `bk.scale(., forward=, backward=)` applies its forward factor only
        # (its `backward` is an nn.Module method that autograd never calls, ``_backends/torch.py:113-124``)
        m_a, m_b = a.sum(), b.sum()
        F_a = (rho + (eps / 2) * m_b) + (rho + eps / 2) * (-(-f_ba / rho).exp())
        G_b = (rho + (eps / 2) * m_a) + (rho + eps / 2) * (-(-g_ab / rho).exp())
    else:
        F_a = (rho + eps / 2) * ((-f_aa / rho).exp() - (-f_ba / rho).exp())
        G_b = (rho + eps / 2) * ((-g_bb / rho).exp() - (-g_ab / rho).exp())
    return (a * F_a).sum() + (b * G_b).sum()
