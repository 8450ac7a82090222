"""float64 clouds on the matrix-free backends (``csrc/glhip_api_f64.hip``): the reference keeps the dtype of its inputs there
(``softmin_online_lazytensor`` _legacy/sinkhorn_samples.py:229-290, ``kernel_online`` _legacy/kernel_samples.py:128-137), rounds
1-3 cast them down to fp32.  Kernels against the float64 C oracle at 1e-12 (dense, batched, block-sparse with an empty row block,
D = 1 ... 16, p = 1 / 2, the three kernels, -inf dual values); ``SamplesLoss`` against the reference's float64 runs (golden
vectors) at 1e-9 — loss, gradients, potentials; the two-scale backend against its float64 oracle."""

import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden, relerr
from geomloss_amd import SamplesLoss, hip
from geomloss_amd.cluster import from_matrix
from oracle import oracle_c, oracle_np

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _clouds(seed, N, M, D, B=None):
    rng = np.random.default_rng(seed)
    shp = (lambda n: (n, D)) if B is None else (lambda n: (B, n, D))
    return rng.random(shp(N)), rng.random(shp(M)) * 0.8 + 0.1, rng.standard_normal(shp(M)[:-1])


@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 9, 16, 17, 20, 37])       # D > 16: f64_generic_kernel (round 5)
@pytest.mark.parametrize("p", [2, 1])
def test_f64_softmin_and_gradient_vs_c_oracle(cuda, D, p):
    N, M = 300, 1000
    x, y, h = _clouds(D + p, N, M, D)
    h[::7] = -np.inf                                   # columns without mass
    g = np.random.default_rng(1).standard_normal(N)
    eps = 0.05 if p == 2 else 0.1
    ref = oracle_c.softmin(eps, x, y, h, p)
    refg = oracle_c.softmin_grad_x(eps, x, y, h, g, p)
    xt = _t(x, cuda).requires_grad_(True)
    out = hip.softmin(eps, xt, _t(y, cuda), _t(h, cuda), p=p)
    assert out.dtype == torch.float64 and relerr(out.detach().cpu().numpy(), ref) < 1e-12
    (gx,) = torch.autograd.grad(out, [xt], grad_outputs=_t(g, cuda))
    assert gx.dtype == torch.float64 and relerr(gx.cpu().numpy(), refg) < 1e-11
    # the composed half-step (no fused float64 kernel)
    prev = _t(np.random.default_rng(2).standard_normal(N), cuda)
    step = hip.sinkhorn_step(eps, xt.detach(), _t(y, cuda), _t(h, cuda), None, prev, 0.7, p=p)
    assert step.dtype == torch.float64 and relerr(step.cpu().numpy(), 0.5 * (prev.cpu().numpy() + 0.7 * ref)) < 1e-12


@pytest.mark.parametrize("kind", ["gaussian", "laplacian", "energy"])
@pytest.mark.parametrize("D", [2, 3, 6, 20])
def test_f64_kernel_products_and_gradients_vs_c_oracle(cuda, kind, D):
    B, N, M = 2, 260, 700
    x, y, v = _clouds(D, N, M, D, B=B)
    v = v / M
    x[0, :5] = y[0, :5]                                 # coincident pairs: the clamp of utils.py:61, zero direction
    g = np.random.default_rng(4).standard_normal((B, N))
    blur = 0.3
    ref = np.stack([oracle_c.kconv(kind, x[b], y[b], v[b], blur) for b in range(B)])
    refg = np.stack([oracle_c.kconv_grad_x(kind, x[b], y[b], v[b], g[b], blur) for b in range(B)])
    xt = _t(x, cuda).requires_grad_(True)
    out = hip.kernel_conv(kind, xt, _t(y, cuda), _t(v, cuda), blur)
    assert out.dtype == torch.float64 and relerr(out.detach().cpu().numpy(), ref) < 1e-12
    (gx,) = torch.autograd.grad(out, [xt], grad_outputs=_t(g, cuda))
    assert relerr(gx.cpu().numpy(), refg) < 1e-11


@pytest.mark.parametrize("D", [2, 3, 9, 20])
@pytest.mark.parametrize("p", [2, 1])
def test_f64_fused_half_step_vs_c_oracle(cuda, D, p):
    """``glhip_sinkhorn_step_f64`` (round 6): out = damping * softmin(eps, C, logw + pot / eps), averaged with ``prev`` when given —
    one launch — against the float64 C oracle at 1e-12: dense, batched, block-sparse; -inf log-weights (columns without mass);
    small row counts (64 threads per row) and larger ones."""
    rng = np.random.default_rng(7 * D + p)
    eps, damping = (0.05 if p == 2 else 0.1), 0.8
    for N, M, B in ((40, 700, None), (1500, 300, None), (130, 257, 3)):
        x, y, logw = _clouds(D + N, N, M, D, B)
        logw = np.log(rng.random(logw.shape) / M)
        logw[..., ::9] = -np.inf
        pot = rng.standard_normal(logw.shape) * 0.05
        prev = rng.standard_normal(logw.shape[:-1] + (N,))
        bs = [None] if B is None else range(B)
        sel = (lambda a, b: a if b is None else a[b])
        soft = np.stack([oracle_c.softmin(eps, sel(x, b), sel(y, b), sel(logw, b) + sel(pot, b) / eps, p) for b in bs]).reshape(prev.shape)
        soft0 = np.stack([oracle_c.softmin(eps, sel(x, b), sel(y, b), sel(logw, b), p) for b in bs]).reshape(prev.shape)
        xt, yt, lw, pt, pv = (_t(a, cuda) for a in (x, y, logw, pot, prev))
        got = hip.sinkhorn_step(eps, xt, yt, lw, pt, pv, damping, p=p)
        assert got.dtype == torch.float64 and relerr(got.cpu().numpy(), 0.5 * (prev + damping * soft)) < 1e-12
        got = hip.sinkhorn_step(eps, xt, yt, lw, pt, None, damping, p=p)
        assert relerr(got.cpu().numpy(), damping * soft) < 1e-12
        got = hip.sinkhorn_step(eps, xt, yt, lw, None, None, damping, p=p)
        assert relerr(got.cpu().numpy(), damping * soft0) < 1e-12
    # block-sparse, with a row block that reduces over nothing (+inf there, like the soft-min itself)
    N, M = 230, 260
    x, y, logw = _clouds(3 * D, N, M, D)
    pot, prev = rng.standard_normal(M) * 0.05, rng.standard_normal(N)
    ri = np.array([[0, 100], [100, 140], [140, 230]], np.int32)
    rj = np.array([[0, 90], [90, 200], [200, 260]], np.int32)
    keep = np.array([[1, 0, 1], [0, 0, 0], [1, 1, 0]], bool)
    rg = from_matrix(_t(ri, cuda), _t(rj, cuda), _t(keep, cuda))
    tup = tuple(t.cpu().numpy() for t in (rg.ranges_i, rg.slices_i, rg.redranges_j))
    want = 0.5 * (prev + damping * oracle_c.softmin(eps, x, y, logw + pot / eps, p, ranges=tup))
    got = hip.sinkhorn_step(eps, _t(x, cuda), _t(y, cuda), _t(logw, cuda), _t(pot, cuda), _t(prev, cuda), damping, p=p, ranges=rg).cpu().numpy()
    live = np.isfinite(want)
    assert (~live).sum() == 40 and np.isposinf(got[~live]).all()
    assert relerr(got[live], want[live]) < 1e-12


def test_f64_block_sparse_with_an_empty_row_block(cuda):
    rng = np.random.default_rng(17)
    N, M, D = 900, 1100, 3
    x, y, h = _clouds(23, N, M, D)
    cut_i, cut_j = np.sort(rng.choice(np.arange(1, N), 6, replace=False)), np.sort(rng.choice(np.arange(1, M), 8, replace=False))
    ri = np.stack([np.r_[0, cut_i], np.r_[cut_i, N]], 1).astype(np.int32)
    rj = np.stack([np.r_[0, cut_j], np.r_[cut_j, M]], 1).astype(np.int32)
    keep = rng.random((7, 9)) < 0.4
    keep[0, :], keep[1, :] = False, True
    rg = from_matrix(_t(ri, cuda), _t(rj, cuda), _t(keep, cuda))
    tup = tuple(t.cpu().numpy() for t in (rg.ranges_i, rg.slices_i, rg.redranges_j))
    live = np.ones(N, bool)
    live[ri[0, 0]:ri[0, 1]] = False
    ref = oracle_c.softmin(0.02, x, y, h, 2, ranges=tup)
    out = hip.softmin(0.02, _t(x, cuda), _t(y, cuda), _t(h, cuda), ranges=rg).cpu().numpy()
    assert relerr(out[live], ref[live]) < 1e-12 and np.isinf(out[~live]).all()
    v = rng.random(M) / M
    refk = oracle_c.kconv("gaussian", x, y, v, 0.1, ranges=tup)
    outk = hip.kernel_conv("gaussian", _t(x, cuda), _t(y, cuda), _t(v, cuda), 0.1, ranges=rg).cpu().numpy()
    assert relerr(outk, refk) < 1e-12


@pytest.mark.parametrize("name", golden_cases())
def test_f64_losses_match_the_reference_float64_runs(cuda, name):
    """Every golden case with float64 inputs on backend="online": the reference's own float64 numbers, to 1e-9."""
    rec = load_golden(name)
    a, x, b, y = (torch.from_numpy(rec[k]).double().to(cuda) for k in "axby")
    x.requires_grad_(True)
    a.requires_grad_(True)
    L = SamplesLoss(backend="online", **rec["kwargs"])(a, x, b, y)
    assert L.dtype == torch.float64 and relerr(L.detach().cpu().numpy(), rec["loss_f64"]) < 1e-9
    gx, ga = torch.autograd.grad(L.sum(), [x, a])
    assert gx.dtype == torch.float64 and relerr(gx.cpu().numpy(), rec["gx_f64"]) < 1e-8 and relerr(ga.cpu().numpy(), rec["ga_f64"]) < 1e-8
    F, G = SamplesLoss(backend="online", potentials=True, **rec["kwargs"])(a.detach(), x.detach(), b, y)
    assert F.dtype == torch.float64 and relerr(F.cpu().numpy(), rec["F_f64"]) < 1e-9 and relerr(G.cpu().numpy(), rec["G_f64"]) < 1e-9


def test_f64_multiscale_matches_the_two_scale_oracle(cuda):
    N, M = 1500, 1400
    rng = np.random.default_rng(3)
    x, y = rng.random((N, 3)), rng.random((M, 3)) * 0.5 + 0.4
    a, b = np.full(N, 1 / N), np.full(M, 1 / M)
    (ref, ref_gx), info = oracle_np.sinkhorn_multiscale(a, x, b, y, p=2, blur=0.05, scaling=0.6, truncate=5, return_info=True, grad=True)
    assert 0 < info["kept_fraction"][0] < 1
    xt = _t(x, cuda).requires_grad_(True)
    L = SamplesLoss("sinkhorn", p=2, blur=0.05, scaling=0.6, backend="multiscale")(xt, _t(y, cuda))
    assert L.dtype == torch.float64 and abs(L.item() - ref) / abs(ref) < 1e-8
    (gx,) = torch.autograd.grad(L, [xt])
    assert relerr(gx.cpu().numpy(), ref_gx) < 1e-7


def test_f64_clouds_of_dimension_20_keep_their_dtype_end_to_end(cuda):
    """float64 clouds with D > 16 through ``SamplesLoss`` on the online backend (round-4 advice: they raised GLHIP_EUNSUPPORTED):
    loss and gradient in float64, against the NumPy oracle; and a block-sparse launch of the generic-D float64 kernel."""
    rng = np.random.default_rng(5)
    N, M, D = 250, 320, 20
    x, y = rng.random((N, D)) * 0.5, rng.random((M, D)) * 0.5 + 0.1
    for name, kw in (("sinkhorn", dict(p=2, blur=0.3)), ("gaussian", dict(blur=0.4)), ("energy", dict())):
        xt = _t(x, cuda).requires_grad_(True)
        L = SamplesLoss(name, backend="online", **kw)(xt, _t(y, cuda))
        (gx,) = torch.autograd.grad(L, [xt])
        assert L.dtype == torch.float64 and gx.dtype == torch.float64
        if name == "sinkhorn":
            ref, rgx, _ = oracle_np.sinkhorn_loss_and_grad(x, y, **kw)
        else:
            ref, rgx = oracle_np.kernel_loss(name, x, y, **kw), oracle_np.kernel_loss_grad_x(name, x, y, **kw)
        assert abs(L.item() - ref) < 1e-9 * abs(ref), name
        assert relerr(gx.cpu().numpy(), rgx) < 1e-8, name
    # block-sparse, D = 20: two row blocks, the first keeps both column blocks, the second only the last
    h = rng.standard_normal(M)
    ri = np.array([[0, 100], [100, N]], np.int32)
    rj = np.array([[0, 150], [150, M]], np.int32)
    keep = np.array([[True, True], [False, True]])
    rg = from_matrix(_t(ri, cuda), _t(rj, cuda), _t(keep, cuda))
    out = hip.softmin(0.1, _t(x, cuda), _t(y, cuda), _t(h, cuda), p=2, ranges=rg).cpu().numpy()
    assert relerr(out[:100], oracle_c.softmin(0.1, x[:100], y, h, 2)) < 1e-12
    assert relerr(out[100:], oracle_c.softmin(0.1, x[100:], y[150:], h[150:], 2)) < 1e-12
