"""Full-size oracle: chunked float64 PyTorch restatement of the reductions of geomloss's hot path.

TEST INFRASTRUCTURE ONLY (same rule as ``oracle_np.py``): imported by ``tests/`` and never by
``geomloss_amd/``.  It exists because the NumPy / C oracles need minutes per soft-min from N = 1e5 on,
while the BASELINE configs are 1e5 ... 1e6 points: here every reduction is evaluated in float64 with
plain ``torch`` tensor operations (``addmm`` / ``logsumexp`` / ``softmax`` on row chunks of the implicit
N x M matrix), on whatever device the caller names — the GPU of the test box for the full sizes, the CPU
for the cross-checks of ``tests/test_oracle_golden.py``.  No HIP kernel of ``geomloss_amd`` is involved.

Restated (paths under /root/reference/src/geomloss/_legacy/):

* ``softmin``            softmin_tensorized, sinkhorn_samples.py:70-71, on the costs of :26-29
                         (utils.py:26-61: |x-y|^2/2 through |x|^2 - 2 x.y + |y|^2 — the reference's own
                         expansion, harmless in float64 on centred clouds — and sqrt(clamp_min(.,1e-8))).
* ``softmin_grad_x``     what autograd returns through that expression for d/dx sum_i g_i f_i.
* ``kconv`` / ``kconv_grad_x``  K @ v with the kernels of kernel_samples.py:62-82, and its x-gradient.
* ``sinkhorn_loss``      sinkhorn_online / sinkhorn_tensorized (:74-221, :349-424): the loop itself is
                         ``oracle_np.sinkhorn_loop`` (pinned against the reference's golden outputs), run
                         with this module's soft-min as its plug-in; closed-form gradient of Appendix A.
* ``kernel_loss``        kernel_loss, kernel_samples.py:92-146, with the closed-form gradient.
* ``sinkhorn_multiscale`` the two-scale algorithm (sinkhorn_samples.py:453-681) at full size: same statements as
                         ``oracle_np.sinkhorn_multiscale`` (whose clustering, loop and truncation rule it reuses), with the
                         fine-level block-sparse reductions evaluated cluster by cluster in float64 on the device
                         instead of on dense masked matrices.

Pinning: ``tests/test_oracle_golden.py::test_torch64_*`` checks every function here against
``oracle_np`` (<= 1e-11) and against the reference-generated golden vectors (<= 1e-8 / 1e-7).
"""

import numpy as np
import torch

from . import oracle_hip64, oracle_np

F64 = torch.float64
# Reductions of >= HIP64_MIN_PAIRS pairs on a GPU go to the brute-force float64 HIP kernels of oracle_hip64.hip (explicit
# differences, one thread per row: seconds where the chunked torch ops below need minutes).  tests/test_full_size_gpu.py pins
# those kernels against this module's torch path and against oracle_c.c on the same box before relying on them.
USE_HIP64 = True
HIP64_MIN_PAIRS = 2e10


def _hip64(device, N, M, D):
    return (USE_HIP64 and torch.device(device).type == "cuda" and float(N) * M >= HIP64_MIN_PAIRS and D <= 16
            and oracle_hip64.available())

_BUDGET = 1 << 26   # matrix entries per chunk (512 MB in float64; a handful of temporaries live at once)


def default_device():
    return torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")


def _t(a, device):
    if isinstance(a, torch.Tensor):
        return a.detach().to(device=device, dtype=F64)
    return torch.as_tensor(np.asarray(a, np.float64), device=device)


def _np(t):
    return t.detach().cpu().numpy()


def _host64(a):
    return a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)


def _chunks(N, M, budget):
    rows = max(1, min(N, budget // max(M, 1)))
    return [(i, min(N, i + rows)) for i in range(0, N, rows)]


def _exponent(xb, y, y2, h, inv_eps, p, exact):
    """(R, M) matrix h_j - C(x_i, y_j)/eps (without the -|x_i|^2/(2 eps) row term when p = 2 and not exact; it is
    returned separately so that one addmm builds the whole matrix)."""
    if p == 2 and not exact:
        # h_j - |y_j|^2/(2 eps) + x_i.y_j/eps
        return torch.addmm((h - 0.5 * inv_eps * y2).unsqueeze(0), xb, y.t(), alpha=inv_eps), -0.5 * inv_eps * (xb * xb).sum(1)
    d2 = torch.zeros((xb.shape[0], y.shape[0]), dtype=F64, device=xb.device)
    for d in range(xb.shape[1]):
        diff = xb[:, d:d + 1] - y[:, d].unsqueeze(0)
        d2 += diff * diff
    C = 0.5 * d2 if p == 2 else torch.sqrt(d2.clamp_min(1e-8))   # utils.py:61
    return h.unsqueeze(0) - inv_eps * C, None


def _centred(x, y):
    c = 0.5 * (x.mean(0, keepdim=True) + y.mean(0, keepdim=True)) if x.shape[0] and y.shape[0] else 0.0
    return x - c, y - c


def softmin(eps, x, y, h, p=2, device=None, exact=False, rows=None, budget=_BUDGET):
    """f_i = -eps log sum_j exp(h_j - C(x_i,y_j)/eps) for x (N,D), y (M,D), h (M,) -> float64 NumPy (N,) (or the rows
    ``rows`` only).  ``exact``: explicit differences instead of the expanded squared distance."""
    device = default_device() if device is None else device
    x, y, h = _t(x, device), _t(y, device), _t(h, device).reshape(-1)
    if rows is not None:
        x = x[torch.as_tensor(rows, device=device)]
    if _hip64(device, x.shape[0], y.shape[0], x.shape[1]):
        return _np(oracle_hip64.softmin(eps, x, y, h, p, device=device))
    x, y = _centred(x, y)
    y2 = (y * y).sum(1)
    out = torch.empty(x.shape[0], dtype=F64, device=device)
    for i0, i1 in _chunks(x.shape[0], y.shape[0], budget):
        E, row = _exponent(x[i0:i1], y, y2, h, 1.0 / eps, p, exact)
        lse = torch.logsumexp(E, dim=1)
        out[i0:i1] = -eps * (lse if row is None else lse + row)
    return _np(out)


def softmin_grad_x(eps, x, y, h, g, p=2, device=None, rows=None, budget=_BUDGET):
    """d/dx sum_i g_i f_i = g_i sum_j P_ij dC/dx(x_i, y_j), P = softmax_j(h_j - C_ij/eps) -> (N,D) (or the rows ``rows``)."""
    device = default_device() if device is None else device
    x, y, h, g = _t(x, device), _t(y, device), _t(h, device).reshape(-1), _t(g, device).reshape(-1)
    if rows is not None:
        sel = torch.as_tensor(rows, device=device)
        x, g = x[sel], g[sel]
    if _hip64(device, x.shape[0], y.shape[0], x.shape[1]):
        return _np(oracle_hip64.softmin_grad_x(eps, x, y, h, g, p, device=device))
    x, y = _centred(x, y)
    y2 = (y * y).sum(1)
    out = torch.empty_like(x)
    for i0, i1 in _chunks(x.shape[0], y.shape[0], budget // 2):
        xb = x[i0:i1]
        E, _ = _exponent(xb, y, y2, h, 1.0 / eps, p, exact=(p != 2))
        P = torch.softmax(E, dim=1)                      # the row term cancels in the soft-max
        if p == 2:                                       # sum_j P_ij (x_i - y_j) = x_i - (P y)_i
            out[i0:i1] = g[i0:i1, None] * (xb - P @ y)
        else:                                            # unit directions, zero where the clamp of utils.py:61 is active
            acc = torch.zeros_like(xb)
            d2 = torch.zeros_like(P)
            for d in range(xb.shape[1]):
                diff = xb[:, d:d + 1] - y[:, d].unsqueeze(0)
                d2 += diff * diff
            inv = torch.where(d2 > 1e-8, d2.clamp_min(1e-300).rsqrt(), torch.zeros_like(d2))
            W = P * inv
            for d in range(xb.shape[1]):
                acc[:, d] = (W * (xb[:, d:d + 1] - y[:, d].unsqueeze(0))).sum(1)
            out[i0:i1] = g[i0:i1, None] * acc
    return _np(out)


def _kernel_rows(kind, xb, y, blur):
    """(R, M) kernel values and, for the gradient, the (R, M) factor c_ij with dk/dx_i = c_ij (x_i - y_j)."""
    if kind == "gaussian":          # exp(-|x-y|^2 / (2 blur^2)), kernel_samples.py:62-68
        # one addmm instead of 3 D elementwise passes: -|x-y|^2/2b^2 = x.y/b^2 - |x|^2/2b^2 - |y|^2/2b^2, the reference's own
        # expansion (utils.py:42-49); the callers centre the clouds, so the cancellation costs ~1e-16 diam^2 / blur^2
        ib2 = 1.0 / (blur * blur)
        K = torch.addmm((-0.5 * ib2 * (y * y).sum(1)).unsqueeze(0), xb, y.t(), alpha=ib2)
        K += (-0.5 * ib2 * (xb * xb).sum(1)).unsqueeze(1)
        K.clamp_max_(0.0).exp_()
        return K, -ib2
    d2 = torch.zeros((xb.shape[0], y.shape[0]), dtype=F64, device=xb.device)
    for d in range(xb.shape[1]):
        diff = xb[:, d:d + 1] - y[:, d].unsqueeze(0)
        d2 += diff * diff
    if kind == "laplacian":         # exp(-sqrt(clamp_min(|x/blur - y/blur|^2, 1e-8))), :71-77
        s2 = d2 / (blur * blur)
        live = s2 > 1e-8
        dist = torch.sqrt(s2.clamp_min(1e-8))
        K = torch.exp(-dist)
        return K, torch.where(live, -K / (dist * blur * blur), torch.zeros_like(K))
    if kind == "energy":            # -sqrt(clamp_min(|x-y|^2, 1e-8)), :80-82
        live = d2 > 1e-8
        dist = torch.sqrt(d2.clamp_min(1e-8))
        return -dist, torch.where(live, -1.0 / dist, torch.zeros_like(dist))
    raise KeyError(kind)


def kconv(kind, x, y, v, blur=0.05, device=None, rows=None, budget=_BUDGET):
    """(K_xy v)_i -> float64 NumPy (N,)."""
    device = default_device() if device is None else device
    x, y, v = _t(x, device), _t(y, device), _t(v, device).reshape(-1)
    if rows is not None:
        x = x[torch.as_tensor(rows, device=device)]
    if _hip64(device, x.shape[0], y.shape[0], x.shape[1]):
        return _np(oracle_hip64.kconv(kind, x, y, v, blur, device=device))
    x, y = _centred(x, y)
    out = torch.empty(x.shape[0], dtype=F64, device=device)
    for i0, i1 in _chunks(x.shape[0], y.shape[0], budget // 2):
        K, _ = _kernel_rows(kind, x[i0:i1], y, blur)
        out[i0:i1] = K @ v
    return _np(out)


def kconv_grad_x(kind, x, y, v, g, blur=0.05, device=None, rows=None, budget=_BUDGET):
    """d/dx sum_i g_i (K_xy v)_i -> (N,D)."""
    device = default_device() if device is None else device
    x, y, v, g = _t(x, device), _t(y, device), _t(v, device).reshape(-1), _t(g, device).reshape(-1)
    if rows is not None:
        sel = torch.as_tensor(rows, device=device)
        x, g = x[sel], g[sel]
    if _hip64(device, x.shape[0], y.shape[0], x.shape[1]):
        return _np(oracle_hip64.kconv(kind, x, y, v, blur, g=g, device=device, value=False)[1])
    x, y = _centred(x, y)
    out = torch.empty_like(x)
    for i0, i1 in _chunks(x.shape[0], y.shape[0], budget // 2):
        xb = x[i0:i1]
        K, c = _kernel_rows(kind, xb, y, blur)
        cv = K.mul_(c).mul_(v.unsqueeze(0)) if kind == "gaussian" else c * v.unsqueeze(0)     # gaussian: c = -K / blur^2
        out[i0:i1] = g[i0:i1, None] * (xb * cv.sum(1, keepdim=True) - cv @ y)   # sum_j c_ij v_j (x_i - y_j)
    return _np(out)


# --------------------------------------------------------------------------------------------------
#  whole losses
# --------------------------------------------------------------------------------------------------


def _weights(n, w):
    return np.full(n, 1.0 / n) if w is None else _host64(w).reshape(-1)


def sinkhorn_loss(x, y, a=None, b=None, p=2, blur=0.05, reach=None, diameter=None, scaling=0.5, debias=True,
                  potentials=False, grad=False, full=False, device=None):
    """SamplesLoss("sinkhorn", backend="online")(a, x, b, y) for ONE pair of clouds (N,D), (M,D), in float64.

    Returns the loss, or ``(F, G)`` with ``potentials``, or ``(loss, dL/dx, dL/da)`` with ``grad`` (balanced only; closed
    form of SURVEY Appendix A, the same as ``oracle_np.sinkhorn_loss_and_grad``), or with ``full`` a dict of all of these
    plus the four raw dual potentials, from ONE run of the loop."""
    device = default_device() if device is None else device
    xn, yn = _host64(x), _host64(y)
    a, b = _weights(xn.shape[0], a), _weights(yn.shape[0], b)
    _, eps, eps_list, rho = oracle_np.scaling_parameters(xn, yn, p, blur, reach, diameter, scaling)
    xt, yt = _t(xn, device), _t(yn, device)

    def sm(eps_, C, h):
        return softmin(eps_, C[0], C[1], h, p=p, device=device)

    C_xx, C_yy = ([(xt, xt)], [(yt, yt)]) if debias else (None, None)
    pots, last = oracle_np.sinkhorn_loop(sm, [oracle_np.log_weights(a)], [oracle_np.log_weights(b)], C_xx, C_yy,
                                         [(xt, yt)], [(yt, xt)], eps_list, rho, debias=debias)
    f_aa, g_bb, g_ab, f_ba = pots
    out = oracle_np.sinkhorn_cost(eps, rho, a, b, f_aa, g_bb, g_ab, f_ba, debias=debias, potentials=potentials and not full)
    if not grad and not full:
        return out if potentials else float(out)
    assert reach is None and (full or not potentials), "closed-form gradient restated for the balanced loss only"
    gx = softmin_grad_x(eps, xt, yt, last["h_ba"], a, p=p, device=device)
    ga = f_ba.copy()
    if debias:
        gx = gx - softmin_grad_x(eps, xt, xt, last["h_aa"], a, p=p, device=device)
        ga = ga - f_aa
    if full:
        F, G = oracle_np.sinkhorn_cost(eps, rho, a, b, f_aa, g_bb, g_ab, f_ba, debias=debias, potentials=True)
        return dict(loss=float(out), gx=gx, ga=ga, F=F, G=G, f_aa=f_aa, g_bb=g_bb, g_ab=g_ab, f_ba=f_ba)
    return float(out), gx, ga


def kernel_loss(name, x, y, a=None, b=None, blur=0.05, potentials=False, grad=False, device=None, budget=_BUDGET):
    """SamplesLoss(name, backend="online")(a, x, b, y) for one pair of clouds; with ``grad``: (loss, dL/dx, dL/da)."""
    device = default_device() if device is None else device
    xn, yn = _host64(x), _host64(y)
    a, b = _weights(xn.shape[0], a), _weights(yn.shape[0], b)
    xt, yt = _t(xn, device), _t(yn, device)
    a_x = kconv(name, xt, xt, a, blur, device, budget=budget)
    b_y = kconv(name, yt, yt, b, blur, device, budget=budget)
    b_x = kconv(name, xt, yt, b, blur, device, budget=budget)
    if potentials:
        return a_x - b_x, b_y - kconv(name, yt, xt, a, blur, device, budget=budget)
    loss = float(0.5 * a @ a_x + 0.5 * b @ b_y - a @ b_x)
    if not grad:
        return loss
    # DoubleGrad (kernel_samples.py:43-54) doubles the half of the symmetric term that autograd sees
    gx = kconv_grad_x(name, xt, xt, a, a, blur, device, budget=budget) - kconv_grad_x(name, xt, yt, b, a, blur, device, budget=budget)
    return loss, gx, a_x - b_x


# --------------------------------------------------------------------------------------------------
#  two-scale Sinkhorn at full size
# --------------------------------------------------------------------------------------------------


_FINE_BUDGET = 1 << 27   # matrix entries per fine-level chunk (1 GB in float64)


def _plan_groups(keep, nx, ny, budget):
    """Runs [k0, k1) of consecutive row clusters whose rows x (union of their kept columns) fit the budget.  Clusters that are
    neighbours in the lexicographic voxel order keep nearly the same column clusters, so a run costs little more than its
    members one by one (the pairs outside a member's own keep set are masked out again: same pair set as the reference)."""
    groups, k, C = [], 0, keep.shape[0]
    while k < C:
        u, rows, k1 = keep[k].copy(), int(nx[k]), k + 1
        while k1 < C:
            u2, rows2 = u | keep[k1], rows + int(nx[k1])
            if rows2 * int(ny[u2].sum()) > budget:
                break
            u, rows, k1 = u2, rows2, k1 + 1
        groups.append((k, k1))
        k = k1
    return groups


class _FineCost:
    """Fine-level cost object: cluster-sorted clouds (device, float64) + an optional cluster-level keep mask."""

    def __init__(self, x, y, ranges_x, ranges_y, keep=None):
        self.x, self.y, self.ranges_x, self.ranges_y = x, y, np.asarray(ranges_x), np.asarray(ranges_y)
        self.keep = None
        if keep is not None:
            dev = x.device
            nx, ny = self.ranges_x[:, 1] - self.ranges_x[:, 0], self.ranges_y[:, 1] - self.ranges_y[:, 0]
            self.keep = torch.as_tensor(np.ascontiguousarray(keep), device=dev)
            self.row_label = torch.repeat_interleave(torch.arange(len(nx), device=dev), torch.as_tensor(nx, device=dev))
            self.col_label = torch.repeat_interleave(torch.arange(len(ny), device=dev), torch.as_tensor(ny, device=dev))
            self.y2 = (y * y).sum(1)
            self.hip64 = _hip64(dev, x.shape[0], y.shape[0], x.shape[1])
            if self.hip64:     # the kept blocks as CSR lists of column intervals for oracle_hip64.hip
                self.pattern = oracle_hip64.make_pattern(keep, self.ranges_x, self.ranges_y, dev)
            else:
                self.groups = _plan_groups(np.asarray(keep, bool), nx, ny, _FINE_BUDGET)


def _fine_reduce(eps, Cobj, h, p, g=None):
    """Block-sparse reduction over the kept (row cluster, column cluster) pairs, a run of row clusters at a time, all on the
    device: soft-min values (N,) or, with ``g``, the gradient g_i sum_j P_ij dC/dx (N, D).  Rows without a kept column:
    +inf (the soft-min of the empty set) / zero gradient."""
    x, y = Cobj.x, Cobj.y
    if Cobj.hip64:
        if g is None:
            return oracle_hip64.softmin(eps, x, y, h, p, pattern=Cobj.pattern, device=x.device)
        return oracle_hip64.softmin_grad_x(eps, x, y, h, g, p, pattern=Cobj.pattern, device=x.device)
    out = torch.empty(x.shape[0] if g is None else x.shape, dtype=F64, device=x.device)
    inv_eps = 1.0 / eps
    for k0, k1 in Cobj.groups:
        r0, r1 = int(Cobj.ranges_x[k0, 0]), int(Cobj.ranges_x[k1 - 1, 1])
        if r1 <= r0:
            continue
        cols = Cobj.keep[k0:k1].any(0)[Cobj.col_label].nonzero().view(-1)
        if cols.numel() == 0:
            out[r0:r1] = float("inf") if g is None else 0.0
            continue
        yc, hc, y2c, cl = y[cols], h[cols], Cobj.y2[cols], Cobj.col_label[cols]
        for i0, i1 in _chunks(r1 - r0, cols.numel(), _FINE_BUDGET):
            i0, i1 = r0 + i0, r0 + i1
            xb = x[i0:i1]
            E, row = _exponent(xb, yc, y2c, hc, inv_eps, p, exact=(p != 2))
            E.masked_fill_(~Cobj.keep[Cobj.row_label[i0:i1]][:, cl], float("-inf"))
            if g is None:
                lse = torch.logsumexp(E, dim=1)
                out[i0:i1] = -eps * (lse if row is None else lse + row)
                continue
            P = torch.nan_to_num(torch.softmax(E, dim=1), nan=0.0)       # a row with no kept column has no plan
            if p == 2:
                out[i0:i1] = g[i0:i1, None] * (xb * P.sum(1, keepdim=True) - P @ yc)
            else:
                acc = torch.zeros_like(xb)
                d2 = torch.zeros_like(P)
                for d in range(xb.shape[1]):
                    diff = xb[:, d:d + 1] - yc[:, d].unsqueeze(0)
                    d2 += diff * diff
                W = P * torch.where(d2 > 1e-8, d2.clamp_min(1e-300).rsqrt(), torch.zeros_like(d2))
                for d in range(xb.shape[1]):
                    acc[:, d] = (W * (xb[:, d:d + 1] - yc[:, d].unsqueeze(0))).sum(1)
                out[i0:i1] = g[i0:i1, None] * acc
    return out


def _softmin_obj(eps, Cobj, h, p, device):
    if isinstance(Cobj, dict):                      # coarse level: dense NumPy matrices of oracle_np
        return oracle_np.softmin_dense(eps, Cobj["C"], h)
    if Cobj.keep is None:
        return softmin(eps, Cobj.x, Cobj.y, h, p=p, device=device)
    return _np(_fine_reduce(eps, Cobj, _t(h, device).reshape(-1), p))


def _softmin_grad_obj(eps, Cobj, h, g, p, device):
    if Cobj.keep is None:
        return softmin_grad_x(eps, Cobj.x, Cobj.y, h, g, p=p, device=device)
    return _np(_fine_reduce(eps, Cobj, _t(h, device).reshape(-1), p, g=_t(g, device).reshape(-1)))


def sinkhorn_multiscale(a, x, b, y, p=2, blur=0.05, reach=None, diameter=None, scaling=0.5, truncate=5, cluster_scale=None,
                        debias=True, potentials=False, grad=False, full=False, device=None, return_info=False,
                        borderline_keeps=None, borderline_tol=1e-6):
    """``SamplesLoss("sinkhorn", backend="multiscale")`` for one pair of clouds at any size: the coarse level on dense NumPy
    matrices exactly as ``oracle_np.sinkhorn_multiscale`` (C ~ 2e3 clusters), the fine level on the device, a run of row
    clusters at a time (``_fine_reduce``).  ``full``: loss, dL/dx and the potentials (caller's point order) from ONE run of the
    loop, as a dict.

    ``borderline_keeps``: the cluster-level keep masks another implementation of the same algorithm decided, in the order the
    loop truncates (xy, xx, yy).  The keep rule `f_i + g_j > C_ij - truncate eps` (sinkhorn_samples.py:512-514) is a threshold on
    quantities two runs in different precisions agree on to ~1e-7 only, so among the ~5e6 cluster pairs of a 1e6-point problem
    a handful sit within rounding of the threshold and are decided differently by a float32 and a float64 run of the SAME code
    (the reference's own included) — each such pair moves the potentials of one cluster by ~eps e^-5 / (kept clusters) ~ 1e-7,
    which is what dominated dL/dx of same-law clouds at N = 1e6 (profiles/r06_full_size_parity.txt).  With the masks given, the
    oracle takes the OTHER side's decision for the pairs where the two differ — after asserting that every one of them is
    borderline, |f_i + g_j - C_ij + truncate eps| <= ``borderline_tol`` — and counts them in ``info["borderline"]``."""
    device = default_device() if device is None else device
    # voxel bins in the precision of the caller's clouds, as the reference evaluates them (oracle_np.grid_cluster)
    label_dtype = np.float32 if isinstance(x, torch.Tensor) and x.dtype == torch.float32 else None
    a, x, b, y = (_host64(t) for t in (a, x, b, y))
    N, D = x.shape
    diameter, eps, eps_list, rho = oracle_np.scaling_parameters(x, y, p, blur, reach, diameter, scaling)
    if cluster_scale is None:
        cluster_scale = diameter / (np.sqrt(D) * 2000 ** (1 / D))
    a_c, a, x_c, x, ranges_x, perm_x = oracle_np.clusterize(a, x, cluster_scale, label_dtype)
    b_c, b, y_c, y, ranges_y, perm_y = oracle_np.clusterize(b, y, cluster_scale, label_dtype)
    jumps, eps_cost = [len(eps_list) - 1], eps
    for i, e in enumerate(eps_list[2:]):
        eps_cost = e                                   # the reference's shadowed `eps` (sinkhorn_samples.py:593-597)
        if cluster_scale**p > e:
            jumps = [i + 1]
            break
    # one common frame for everything that reaches the device: the expanded squared distance of _exponent then acts on offsets
    # of at most one diameter (float64: ~1e-16 diam^2 / eps on an exponent)
    centre = 0.5 * (x.mean(0, keepdims=True) + y.mean(0, keepdims=True))
    xt, yt = _t(x - centre, device), _t(y - centre, device)
    info = dict(jumps=jumps, n_clusters=(len(x_c), len(y_c)), kept_fraction=[], eps_list=eps_list, borderline=[])

    def coarse(u, v, ru, rv):
        return dict(C=oracle_np.cost_matrix(u, v, p), x=u, y=v, ranges_x=ru, ranges_y=rv)

    def sm(eps_, Cobj, h):
        return _softmin_obj(eps_, Cobj, h, p, device)

    def kernel_truncation(C_xy, C_yx, C_xy_f, C_yx_f, f, g, eps_, truncate=None, cost=None):
        if truncate is None:
            return C_xy_f, C_yx_f
        slack = f[:, None] + g[None, :] - (C_xy["C"] - truncate * eps_)
        keep = slack > 0                                                         # sinkhorn_samples.py:512-514
        if borderline_keeps is not None:
            other = borderline_keeps[len(info["kept_fraction"])]
            other = keep if other is None else np.asarray(other, bool)      # None: the other side reduced this term densely
            assert other.shape == keep.shape, (other.shape, keep.shape)
            differ = other != keep
            worst = float(np.abs(slack[differ]).max()) if differ.any() else 0.0
            assert worst <= borderline_tol, f"{int(differ.sum())} keep decisions differ, one by a slack of {worst:.3e}: not a rounding matter"
            info["borderline"].append((int(differ.sum()), worst))
            keep = other
        ni, nj = np.diff(C_xy["ranges_x"], axis=1)[:, 0], np.diff(C_xy["ranges_y"], axis=1)[:, 0]
        info["kept_fraction"].append(float((ni[:, None] * nj[None, :] * keep).sum() / (ni.sum() * nj.sum())))
        return (_FineCost(C_xy_f.x, C_xy_f.y, C_xy["ranges_x"], C_xy["ranges_y"], keep),
                _FineCost(C_yx_f.x, C_yx_f.y, C_xy["ranges_y"], C_xy["ranges_x"], keep.T.copy()))

    extrapolations = []

    def extrapolate(f, g, eps_, lam, C_xy, b_log, C_xy_f):                       # :533-544
        h = b_log + g / eps_
        yc = _t(C_xy["y"] - centre, device)                                      # coarse cloud, in the frame of the fine one
        extrapolations.append((eps_, C_xy_f.x, yc, h))
        return lam * softmin(eps_, C_xy_f.x, yc, h, p=p, device=device)

    C_xys = [coarse(x_c, y_c, ranges_x, ranges_y), _FineCost(xt, yt, ranges_x, ranges_y)]
    C_yxs = [coarse(y_c, x_c, ranges_y, ranges_x), _FineCost(yt, xt, ranges_y, ranges_x)]
    C_xxs = [coarse(x_c, x_c, ranges_x, ranges_x), _FineCost(xt, xt, ranges_x, ranges_x)] if debias else None
    C_yys = [coarse(y_c, y_c, ranges_y, ranges_y), _FineCost(yt, yt, ranges_y, ranges_y)] if debias else None
    lw = oracle_np.log_weights
    pots, last = oracle_np.sinkhorn_loop(sm, [lw(a_c), lw(a)], [lw(b_c), lw(b)], C_xxs, C_yys, C_xys, C_yxs, eps_list, rho,
                                         jumps=jumps, kernel_truncation=kernel_truncation, truncate=truncate,
                                         extrapolate=extrapolate, debias=debias)
    f_aa, g_bb, g_ab, f_ba = pots
    out = oracle_np.sinkhorn_cost(eps_cost, rho, a, b, f_aa, g_bb, g_ab, f_ba, debias=debias, potentials=potentials and not full)

    def unsort(F, G):
        f_x, g_y = np.empty_like(F), np.empty_like(G)
        f_x[perm_x], g_y[perm_y] = F, G
        return f_x, g_y

    if potentials and not full:
        out = unsort(*out)
    else:
        out = float(out)
    if grad or full:
        assert rho is None and (full or not potentials)
        if "h_ba" in last:
            gs = _softmin_grad_obj(last["eps"], last["C_xy"], last["h_ba"], a, p, device)
            if debias:
                gs = gs - _softmin_grad_obj(last["eps"], last["C_xx"], last["h_aa"], a, p, device)
        else:
            e, xf, yc, h = extrapolations[0]
            gs = softmin_grad_x(e, xf, yc, h, a, p=p, device=device)
            if debias:
                e, xf, xc, h = extrapolations[2]
                gs = gs - softmin_grad_x(e, xf, xc, h, a, p=p, device=device)
        gx = np.empty_like(gs)
        gx[perm_x] = gs
        if full:
            F, G = unsort(*oracle_np.sinkhorn_cost(eps_cost, rho, a, b, f_aa, g_bb, g_ab, f_ba, debias=debias, potentials=True))
            raw = max(np.abs(f_ba).max(), np.abs(g_ab).max())                    # scale of the raw dual values
            out = dict(loss=out, gx=gx, F=F, G=G, dual_scale=float(raw), info=info)
        else:
            out = (out, gx)
    return (out, info) if return_info else out
