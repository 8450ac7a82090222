"""PyTorch-CPU port of the reference's *tensorized* Sinkhorn path, used as the timed CPU baseline.

TEST / BENCH INFRASTRUCTURE ONLY (``bench.py`` ``cpu_baseline`` leg, kind = "port").  The reference itself
is Python and cannot travel to the GPU box, so this file restates, op for op, what
``SamplesLoss("sinkhorn", backend="tensorized")`` executes on CPU tensors: dense costs through
|x|^2 - 2 x.y + |y|^2 (``_legacy/utils.py:39-53``), ``logsumexp`` soft-mins
(``_legacy/sinkhorn_samples.py:70-71``) and the loop of ``_legacy/sinkhorn_divergence.py:434-628``.
Checked against the golden vectors in tests/test_oracle_golden.py::test_torch_port_matches_reference.
"""

import numpy as np
import torch


def _cost(x, y, p):
    d2 = (x * x).sum(-1)[:, :, None] - 2 * torch.matmul(x, y.transpose(1, 2)) + (y * y).sum(-1)[:, None, :]
    return d2 / 2 if p == 2 else torch.sqrt(torch.clamp_min(d2, 1e-8))


def _softmin(eps, C, h):
    return -eps * (h[:, None, :] - C / eps).logsumexp(2)


def sinkhorn_tensorized_cpu(x, y, p=2, blur=0.05, scaling=0.5, diameter=None, count=None):
    """Uniform weights, balanced, debiased.  x (B,N,D), y (B,M,D) CPU tensors -> (B,) losses.
    ``count``: optional dict receiving the number of soft-min calls (for pairs/s)."""
    B, N, _ = x.shape
    M = y.shape[1]
    a_log = torch.full((B, N), -float(np.log(N)), dtype=x.dtype)
    b_log = torch.full((B, M), -float(np.log(M)), dtype=x.dtype)
    C_xy, C_yx, C_xx, C_yy = _cost(x, y, p), _cost(y, x, p), _cost(x, x, p), _cost(y, y, p)
    if diameter is None:
        pts = torch.cat((x.reshape(-1, x.shape[-1]), y.reshape(-1, y.shape[-1])))
        diameter = (pts.max(0)[0] - pts.min(0)[0]).norm().item()
    eps_list = ([diameter**p]
                + [np.exp(e) for e in np.arange(p * np.log(diameter), p * np.log(blur), p * np.log(scaling))]
                + [blur**p])
    n_calls = 0
    with torch.no_grad():
        eps = eps_list[0]
        g_ab, f_ba = _softmin(eps, C_yx, a_log), _softmin(eps, C_xy, b_log)
        f_aa, g_bb = _softmin(eps, C_xx, a_log), _softmin(eps, C_yy, b_log)
        n_calls += 4
        for eps in eps_list:
            ft_ba = _softmin(eps, C_xy, b_log + g_ab / eps)
            gt_ab = _softmin(eps, C_yx, a_log + f_ba / eps)
            ft_aa = _softmin(eps, C_xx, a_log + f_aa / eps)
            gt_bb = _softmin(eps, C_yy, b_log + g_bb / eps)
            f_ba, g_ab = 0.5 * (f_ba + ft_ba), 0.5 * (g_ab + gt_ab)
            f_aa, g_bb = 0.5 * (f_aa + ft_aa), 0.5 * (g_bb + gt_bb)
            n_calls += 4
        f_ba, g_ab = _softmin(eps, C_xy, b_log + g_ab / eps), _softmin(eps, C_yx, a_log + f_ba / eps)
        f_aa, g_bb = _softmin(eps, C_xx, a_log + f_aa / eps), _softmin(eps, C_yy, b_log + g_bb / eps)
        n_calls += 4
    if count is not None:
        count["softmin_calls"] = n_calls
    return ((f_ba - f_aa) / N).sum(1) + ((g_ab - g_bb) / M).sum(1)
