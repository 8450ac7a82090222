"""Symmetric, epsilon-scaled Sinkhorn loop and the loss formulas evaluated on its potentials.

Host-side restatement of the reference's solver (``_legacy/sinkhorn_divergence.py``): this part of
the hot path stays in Python on PyTorch tensors and calls the HIP kernels through the ``softmin``
callable it is handed — the reference's own plugin seam (``sinkhorn_loop(softmin, ...)``,
``_legacy/sinkhorn_divergence.py:258``).  Function names, argument order and numerical behaviour
follow the reference so that its call sites work unchanged.
"""

import sys
import types

import numpy as np
import torch

from .utils import scal_sum


class _SolverModule(types.ModuleType):
    """`from geomloss import sinkhorn_divergence` is a *function* in the reference (the image OT solver); here the same name
    is this module, so the module is made callable and forwards to that function."""

    def __call__(self, *args, **kwargs):
        from .sinkhorn_images import sinkhorn_divergence as image_solver

        return image_solver(*args, **kwargs)


sys.modules[__name__].__class__ = _SolverModule


def dampening(eps, rho):
    """Unbalanced-OT damping factor 1 / (1 + eps/rho); 1 when rho is None (``:56-58``)."""
    return 1 if rho is None else 1 / (1 + eps / rho)


def _hip():
    from . import hip      # (imported on first use: the dense CPU paths of this module never touch the library)
    return hip


def _fused_cost_applies(a, b, f_aa, g_bb, g_ab, f_ba, batch):
    """fp32 CUDA potentials and weights of matching shapes, at most 32768 points per measure: (B,N) / (B,M) with ``batch``, vectors without."""
    ts = (a, b, f_aa, g_bb, g_ab, f_ba)
    if not all(t is None or (torch.is_tensor(t) and t.is_cuda) for t in ts) or not _hip().small_ends_apply(*ts):
        return False
    if f_ba.dim() != (2 if batch else 1) or g_ab.dim() != f_ba.dim() or (batch and g_ab.shape[0] != f_ba.shape[0]):
        return False
    B = f_ba.shape[0] if batch else 1
    ok = lambda w, f: w.numel() in (f.shape[-1], B * f.shape[-1]) and w.shape[-1] == f.shape[-1]  # noqa: E731
    same = lambda u, f: u is None or u.shape == f.shape  # noqa: E731
    return ok(a, f_ba) and ok(b, g_ab) and same(f_aa, f_ba) and same(g_bb, g_ab) and B <= 65535


def log_weights(a):
    """log(a), with log(0) replaced by -100000 (``:61-65``)."""
    # clamps instead of the reference's masked assignment: same values, but no boolean-mask indexing (which goes
    # through `nonzero` and a host round trip on every call)
    # (round 5: three launches instead of four — weights <= 0 become log 0 = -inf and then the floor; log a > -104 for every positive
    # float32, so no other value moves; NaN weights stay NaN as in the reference)
    return a.clamp_min(0).log().clamp_min(-100000.0)


def log_weights_many(weights):
    """:func:`log_weights` of several measures, detached (every use of a log-weight vector in the loop is), with three multi-tensor
    launches for the lot where torch offers them (the 4 vectors of a two-scale loss: 3 launches instead of 12)."""
    ws = [w.detach() for w in weights]
    if 1 <= len(ws) <= 4 and _hip().small_ends_apply(*ws):      # a loss on a few thousand points: one launch (glhip_log_weights)
        return _hip().log_weights_raw(ws)
    if len(ws) > 1 and all(w.is_cuda for w in ws) and hasattr(torch, "_foreach_clamp_min_") and hasattr(torch, "_foreach_log_"):
        out = torch._foreach_clamp_min(ws, 0)
        torch._foreach_log_(out)
        torch._foreach_clamp_min_(out, -100000.0)
        return list(out)
    return [log_weights(w) for w in ws]


class UnbalancedWeight(torch.nn.Module):
    """Scales exponentiated potentials by (rho + eps/2) (``:68-88``).

    The reference documents a different backward factor (rho + eps) but implements it as an
    ``nn.Module.backward`` method that autograd never calls; we keep the behaviour that actually
    runs: gradients use the forward factor.
    """

    def __init__(self, eps, rho):
        super().__init__()
        self.eps, self.rho = eps, rho

    def forward(self, x):
        return (self.rho + self.eps / 2) * x

    def backward(self, g):
        return (self.rho + self.eps) * g


def max_diameter(x, y):
    """Length of the diagonal of the joint bounding box of x (N,D) and y (M,D) (``:96-112``)."""
    if x.is_cuda and _hip().bounding_box_applies(x.detach(), y.detach()):      # small clouds: the two reductions as one launch (exact)
        mins, maxs = _hip().bounding_box(x.detach(), y.detach())
    elif x.dtype == y.dtype and x.device == y.device:      # (one reduction over both clouds: 5 launches instead of 9)
        z = torch.cat((x, y))
        if z.is_cuda and z.shape[0] >= 32768:
            # torch reduces dim 0 of a row-major (L, D) tensor at ~30 GB/s (80 us at L = 2e5, 0.8 ms at 2e6): along the contiguous
            # axis of a (D, L) copy instead (exact either way: minima and maxima)
            mins, maxs = torch.aminmax(z.t().contiguous(), dim=1)
        else:
            mins, maxs = torch.aminmax(z, dim=0)
    else:
        (x_min, x_max), (y_min, y_max) = torch.aminmax(x, dim=0), torch.aminmax(y, dim=0)
        mins, maxs = torch.minimum(x_min, y_min), torch.maximum(x_max, y_max)
    return (maxs - mins).norm().item()


def epsilon_schedule(p, diameter, blur, scaling):
    """[diam^p] + geometric ladder from diam^p down to blur^p (ratio scaling^p) + [blur^p] (``:115-151``).

    The ladder is generated by the same ``np.arange`` expression as the reference: its float-step
    rounding decides the number of iterations.
    """
    ladder = np.arange(p * np.log(diameter), p * np.log(blur), p * np.log(scaling))
    return [diameter**p] + [np.exp(e) for e in ladder] + [blur**p]


def scaling_parameters(x, y, p, blur, reach, diameter, scaling):
    """(diameter, eps, eps_list, rho) from the user-level arguments (``:154-163``)."""
    if diameter is None:
        D = x.shape[-1]
        if x.dtype in (torch.bfloat16, torch.float16):  # half-precision clouds: measure the box in fp32
            x, y = x.float(), y.float()
        diameter = max_diameter(x.view(-1, D), y.view(-1, D))
    eps = blur**p
    rho = None if reach is None else reach**p
    return diameter, eps, epsilon_schedule(p, diameter, blur, scaling), rho


def sinkhorn_cost(eps, rho, a, b, f_aa, g_bb, g_ab, f_ba, batch=False, debias=True, potentials=False):
    """Loss value (or dual potentials) from the four potentials (``:171-250``)."""
    if potentials:
        return (f_ba - f_aa, g_ab - g_bb) if debias else (f_ba, g_ab)

    if rho is None:  # balanced
        if _fused_cost_applies(a, b, f_aa if debias else None, g_bb if debias else None, g_ab, f_ba, batch):
            # a loss on a few thousand points is bound by the launch rate: the seven elementwise launches below as one (glhip_sinkhorn_cost)
            return _hip().sinkhorn_cost_fused(a, f_ba, f_aa if debias else None, b, g_ab, g_bb if debias else None, batch)
        if debias:
            return scal_sum(a, f_ba - f_aa, b, g_ab - g_bb, batch=batch)
        return scal_sum(a, f_ba, b, g_ab, batch=batch)

    weight = UnbalancedWeight(eps, rho)  # unbalanced: Sejourne et al. 2019, Prop. 12
    if debias:
        return scal_sum(a, weight((-f_aa / rho).exp() - (-f_ba / rho).exp()),
                        b, weight((-g_bb / rho).exp() - (-g_ab / rho).exp()), batch=batch)
    return scal_sum(a, weight(1 - (-f_ba / rho).exp()), b, weight(1 - (-g_ab / rho).exp()), batch=batch)


def sinkhorn_loop(*args, **kwargs):
    was_enabled = torch.is_grad_enabled()
    try:
        return _sinkhorn_loop(*args, **kwargs)
    except BaseException:
        # the iterations run with autograd switched off by hand (as in the reference): an error raised in there — a CPU tensor
        # handed to a HIP backend, a shape mismatch — must not leave the whole process without gradients
        torch.autograd.set_grad_enabled(was_enabled)
        raise


def _sinkhorn_loop(
    softmin,
    a_logs,
    b_logs,
    C_xxs,
    C_yys,
    C_xys,
    C_yxs,
    eps_list,
    rho,
    jumps=[],
    kernel_truncation=None,
    truncate=5,
    cost=None,
    extrapolate=None,
    debias=True,
    last_extrapolation=True,
):
    """Symmetric Sinkhorn iterations with annealing and optional coarse-to-fine jumps (``:258-628``).

    ``softmin(eps, C, h)`` returns ``-eps * log sum_j exp(h_j - C_ij / eps)`` for an opaque cost
    object ``C``.  Returns ``(f_aa, g_bb, g_ab, f_ba)`` (the first two are None without ``debias``).
    Like the reference, the loop runs with autograd disabled, re-enables it before the last update
    (whose inputs are detached) and leaves it enabled on exit.
    """
    # Optional fast path (not in the reference): a soft-min object may offer `step(eps, C, log_w, pot, damping, prev)`
    # = (prev + damping * softmin(eps, C, log_w + pot/eps)) / 2  (or damping * softmin(...) when prev is None) as ONE
    # kernel launch.  It is only used in the autograd-free part of the loop; values agree to fp32 rounding.
    fused = getattr(softmin, "step", None)

    def first(C, log_w, eps, damping):
        if fused is not None:
            return fused(eps, C, log_w, None, damping, None)
        return damping * softmin(eps, C, log_w)

    def averaged(C, log_w, pot, prev, eps, damping):
        if fused is not None:
            return fused(eps, C, log_w, pot, damping, prev)
        return 0.5 * (prev + damping * softmin(eps, C, log_w + pot / eps))

    # ... and `iter4(eps, C_xy, a_log, b_log, old_potentials | None, damping, debias)` = every simultaneous update of one
    # iteration as ONE launch; it may answer None ("not for this problem"), then the per-soft-min path below runs.
    fused4 = getattr(softmin, "iter4", None)

    multiscale = type(a_logs) is list
    if not multiscale:
        a_logs, b_logs = [a_logs], [b_logs]
        C_xys, C_yxs = [C_xys], [C_yxs]
        if debias:
            C_xxs, C_yys = [C_xxs], [C_yys]

    torch.autograd.set_grad_enabled(False)

    level = 0
    a_log, b_log = a_logs[level], b_logs[level]
    C_xy, C_yx = C_xys[level], C_yxs[level]
    C_xx, C_yy = (C_xxs[level], C_yys[level]) if debias else (None, None)

    n_eps = len(eps_list)
    before_last = None      # the potentials going into the last iteration (for the one-reduction final update)

    # ... and `anneal(eps_list, dampings, C_xy, a_log, b_log, debias)` = the initialisation and EVERY iteration of the first level —
    # all of a single-scale loop, a two-scale one up to its jump — queued by one call; it answers `(final potentials, inputs of the
    # last iteration)` or None.  Same launches as `iter4`, without the interpreter between them.
    done = -1               # index of the last iteration the hook has run
    anneal = getattr(softmin, "anneal", None)
    if anneal is not None and n_eps > 0:
        k = min(min(jumps), n_eps - 1) if (multiscale and len(jumps)) else n_eps - 1
        res = anneal(eps_list[: k + 1], [dampening(e, rho) for e in eps_list[: k + 1]], C_xy, a_log, b_log, debias)
        if res is not None:
            new, old = res
            f_ba, g_ab = new[0], new[1]
            if debias:
                f_aa, g_bb = new[2], new[3]
            done = k
            if k == n_eps - 1:
                before_last = (level, old[0], old[1], old[2] if debias else None, old[3] if debias else None)

    if done < 0:
        # initial potentials at the largest temperature
        eps = eps_list[0]
        damping = dampening(eps, rho)
        init = fused4(eps, C_xy, a_log, b_log, None, damping, debias) if fused4 is not None else None
        if init is not None:
            f_ba, g_ab = init[0], init[1]
            if debias:
                f_aa, g_bb = init[2], init[3]
        else:
            g_ab = first(C_yx, a_log, eps, damping)
            f_ba = first(C_xy, b_log, eps, damping)
            if debias:
                f_aa = first(C_xx, a_log, eps, damping)
                g_bb = first(C_yy, b_log, eps, damping)

    for i, eps in enumerate(eps_list):
        damping = dampening(eps, rho)
        if i > done and i == n_eps - 1:
            before_last = (level, f_ba, g_ab, f_aa if debias else None, g_bb if debias else None)

        if i > done:        # (iterations up to `done` were run by the `anneal` hook above)
            # "simultaneous" updates followed by averaging: every right-hand side uses the OLD potentials
            new = None
            # single scale, the (dense) coarse level of the multiscale scheme, or a fine level that kernel_truncation left dense
            if fused4 is not None and (level == 0 or all(C is None or C[4] is None for C in (C_xy, C_xx, C_yy))):
                new = fused4(eps, C_xy, a_log, b_log, (f_ba, g_ab, f_aa, g_bb) if debias else (f_ba, g_ab), damping, debias)
            if new is not None:
                f_ba, g_ab = new[0], new[1]
                if debias:
                    f_aa, g_bb = new[2], new[3]
            else:
                f_ba, g_ab = (
                    averaged(C_xy, b_log, g_ab, f_ba, eps, damping),
                    averaged(C_yx, a_log, f_ba, g_ab, eps, damping),
                )
                if debias:
                    f_aa, g_bb = (
                        averaged(C_xx, a_log, f_aa, f_aa, eps, damping),
                        averaged(C_yy, b_log, g_bb, g_bb, eps, damping),
                    )

        if i in jumps:  # coarse -> fine
            nxt = level + 1
            if i == n_eps - 1:
                # jump after the last iteration: plain extrapolation, which is then the differentiable step
                C_xy_fine, C_yx_fine = C_xys[nxt], C_yxs[nxt]
                if debias:
                    C_xx_fine, C_yy_fine = C_xxs[nxt], C_yys[nxt]
                last_extrapolation = False
                torch.autograd.set_grad_enabled(True)
            else:
                # optional hook (not in the reference): a kernel_truncation that counts the pairs its rule keeps may do so for the three
                # calls below at once (`prefetch`: one host round trip instead of three)
                prefetch = getattr(kernel_truncation, "prefetch", None)
                kw = [{}, {}, {}]
                if prefetch is not None:
                    calls = [(C_xy, C_yx, f_ba, g_ab)] + ([(C_xx, C_xx, f_aa, f_aa), (C_yy, C_yy, g_bb, g_bb)] if debias else [])
                    kw = [{} if k is None else {"kept": k} for k in prefetch(calls, eps, truncate=truncate, cost=cost)] + [{}, {}]
                C_xy_fine, C_yx_fine = kernel_truncation(
                    C_xy, C_yx, C_xys[nxt], C_yxs[nxt], f_ba, g_ab, eps, truncate=truncate, cost=cost, **kw[0]
                )
                if debias:
                    C_xx_fine, _ = kernel_truncation(
                        C_xx, C_xx, C_xxs[nxt], C_xxs[nxt], f_aa, f_aa, eps, truncate=truncate, cost=cost, **kw[1]
                    )
                    C_yy_fine, _ = kernel_truncation(
                        C_yy, C_yy, C_yys[nxt], C_yys[nxt], g_bb, g_bb, eps, truncate=truncate, cost=cost, **kw[2]
                    )

            # optional hook (not in the reference): `extrapolate.all4(potentials, eps, damping, C_xy, C_yx, a_log, b_log, C_xy_fine,
            # C_yx_fine, debias)` = the four extrapolations below as ONE launch, or None; only where autograd is off
            all4 = getattr(extrapolate, "all4", None) if not torch.is_grad_enabled() else None
            new = None
            if all4 is not None:
                new = all4((f_ba, g_ab, f_aa, g_bb) if debias else (f_ba, g_ab), eps, damping, C_xy, C_yx, a_log, b_log,
                           C_xy_fine, C_yx_fine, debias)
            if new is not None:
                f_ba, g_ab = new[0], new[1]
                if debias:
                    f_aa, g_bb = new[2], new[3]
            else:
                f_ba, g_ab = (
                    extrapolate(f_ba, g_ab, eps, damping, C_xy, b_log, C_xy_fine),
                    extrapolate(g_ab, f_ba, eps, damping, C_yx, a_log, C_yx_fine),
                )
                if debias:
                    f_aa = extrapolate(f_aa, f_aa, eps, damping, C_xx, a_log, C_xx_fine)
                    g_bb = extrapolate(g_bb, g_bb, eps, damping, C_yy, b_log, C_yy_fine)

            level = nxt
            a_log, b_log = a_logs[level], b_logs[level]
            C_xy, C_yx = C_xy_fine, C_yx_fine
            if debias:
                C_xx, C_yy = C_xx_fine, C_yy_fine

    torch.autograd.set_grad_enabled(True)

    # single scale — or a fine level that kernel_truncation left dense, which the fused iterations above ran as a single-scale loop
    # (only a soft-min that offers `last4` has cost objects of that layout: the image pyramids carry other things)
    last4 = getattr(softmin, "last4", None) if last_extrapolation else None
    if last4 is not None and multiscale and not all(C is None or C[4] is None for C in (C_xy, C_xx, C_yy)):
        last4 = None
    fused_last = None
    if last4 is not None:
        fused_last = last4(eps, C_xy, C_yx, a_log, b_log, (f_ba, g_ab, f_aa, g_bb) if debias else (f_ba, g_ab), damping, debias)
    if fused_last is not None:
        f_ba, g_ab = fused_last[0], fused_last[1]
        if debias:
            f_aa, g_bb = fused_last[2], fused_last[3]
    elif last_extrapolation:
        # one non-averaged update; only the first cloud of each cost object carries gradients.  A soft-min object may offer
        # `value_and_grad(eps, C, log_w, pot_new, pot_old, f_new, f_old, damping)`: value and row gradient from one reduction,
        # bounded by the previous iteration (same temperature, same cost); it answers None when that does not apply.
        one_pass = getattr(softmin, "value_and_grad", None)
        same_level = before_last is not None and before_last[0] == level

        def final(C, log_w, pot_new, pot_old, f_new, f_old):
            if one_pass is not None and same_level:
                out = one_pass(eps, C, log_w, pot_new, pot_old, f_new, f_old, damping)
                if out is not None:
                    return out
            return damping * softmin(eps, C, (log_w + pot_new / eps).detach())

        o = before_last if same_level else (None, None, None, None, None)
        f_ba, g_ab = (
            final(C_xy, b_log, g_ab, o[2], f_ba, o[1]),
            final(C_yx, a_log, f_ba, o[1], g_ab, o[2]),
        )
        if debias:
            f_aa = final(C_xx, a_log, f_aa, o[3], f_aa, o[3])
            g_bb = final(C_yy, b_log, g_bb, o[4], g_bb, o[4])

    if debias:
        return f_aa, g_bb, g_ab, f_ba
    return None, None, g_ab, f_ba


sinkhorn_loop.__doc__ = _sinkhorn_loop.__doc__
sinkhorn_loop.__wrapped__ = _sinkhorn_loop      # inspect.signature(sinkhorn_loop) shows the reference's argument list
