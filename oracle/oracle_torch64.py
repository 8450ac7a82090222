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

Pinning: ``tests/test_oracle_golden.py::test_torch64_*`` checks every function here against
``oracle_np`` (<= 1e-11) and against the reference-generated golden vectors (<= 1e-8 / 1e-7).
"""

import numpy as np
import torch

from . import oracle_np

F64 = torch.float64
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
    d2 = torch.zeros((xb.shape[0], y.shape[0]), dtype=F64, device=xb.device)
    for d in range(xb.shape[1]):
        diff = xb[:, d:d + 1] - y[:, d].unsqueeze(0)
        d2 += diff * diff
    if kind == "gaussian":          # exp(-|x-y|^2 / (2 blur^2)), kernel_samples.py:62-68
        K = torch.exp(-d2 / (2 * blur * blur))
        return K, -K / (blur * blur)
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
    out = torch.empty_like(x)
    for i0, i1 in _chunks(x.shape[0], y.shape[0], budget // 2):
        xb = x[i0:i1]
        _, c = _kernel_rows(kind, xb, y, blur)
        cv = c * v.unsqueeze(0)
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


def kernel_loss(name, x, y, a=None, b=None, blur=0.05, potentials=False, grad=False, device=None):
    """SamplesLoss(name, backend="online")(a, x, b, y) for one pair of clouds; with ``grad``: (loss, dL/dx, dL/da)."""
    device = default_device() if device is None else device
    xn, yn = _host64(x), _host64(y)
    a, b = _weights(xn.shape[0], a), _weights(yn.shape[0], b)
    xt, yt = _t(xn, device), _t(yn, device)
    a_x = kconv(name, xt, xt, a, blur, device)
    b_y = kconv(name, yt, yt, b, blur, device)
    b_x = kconv(name, xt, yt, b, blur, device)
    if potentials:
        return a_x - b_x, b_y - kconv(name, yt, xt, a, blur, device)
    loss = float(0.5 * a @ a_x + 0.5 * b @ b_y - a @ b_x)
    if not grad:
        return loss
    # DoubleGrad (kernel_samples.py:43-54) doubles the half of the symmetric term that autograd sees
    gx = kconv_grad_x(name, xt, xt, a, a, blur, device) - kconv_grad_x(name, xt, yt, b, a, blur, device)
    return loss, gx, a_x - b_x
