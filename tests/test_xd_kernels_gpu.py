"""Clouds of dimension 4 <= D <= 16 on the matrix cores (``csrc/glhip_softmin_xd.h``): soft-min forward, fused Sinkhorn half-step
and gaussian kernel product, against the C oracle (``oracle/oracle_c.c``, float64) and — for launches the C oracle would take
minutes on — sampled rows of the chunked float64 oracle (``oracle/oracle_torch64.py``).  The reference takes any D
(``Vi({D})``, _legacy/sinkhorn_samples.py:322-334; its multiscale tutorial clusters 4-D clouds with user labels:
examples/sinkhorn_multiscale/plot_optimal_transport_cluster.py:58-61,155-166).  Everything goes through the C-ABI.
"""

import math

import numpy as np
import pytest
import torch

from conftest import load_golden, relerr
from geomloss_amd import SamplesLoss, hip
from geomloss_amd.cluster import from_matrix
from oracle import oracle_c
from oracle import oracle_torch64 as o64

pytestmark = pytest.mark.gpu

XD = [4, 5, 8, 9, 16]


def _clouds(seed, N, M, D, B=None):
    rng = np.random.default_rng(seed)
    shp = (lambda n: (n, D)) if B is None else (lambda n: (B, n, D))
    x = rng.random(shp(N)).astype(np.float32)
    y = (rng.random(shp(M)) * 0.8 + 0.1).astype(np.float32)
    h = rng.standard_normal(shp(M)[:-1]).astype(np.float32)
    return x, y, h


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _tol(ref, D):
    return 4e-7 * D + 2e-6 * np.abs(ref).max()       # expanded form in fp32: ~2^-22 diam^2 on a potential; diam^2 <= D here


@pytest.mark.parametrize("D", XD)
@pytest.mark.parametrize("N,M", [(300, 257), (1030, 2100), (64, 8), (1, 1), (5, 3000)])
@pytest.mark.parametrize("eps", [1.0, 0.05**2])
def test_softmin_fwd_vs_oracle(cuda, D, N, M, eps):
    x, y, h = _clouds(N + M + D, N, M, D)
    ref = oracle_c.softmin(eps, x, y, h, 2)
    # matrix cores with / without column splits, on both K layouts (f16 x 2: in range here, diameter^2 / eps <= 6400); the VALU fallback
    for flags in (0, hip.FLAG_NO_SPLIT, hip.FLAG_F16X2, hip.FLAG_F16X2 | hip.FLAG_NO_SPLIT, hip.FLAG_NO_MFMA):
        out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), p=2, flags=flags).cpu().numpy()
        assert np.abs(out - ref).max() < _tol(ref, D), flags


@pytest.mark.parametrize("D,N,M", [(4, 700, 70_001), (16, 300, 66_000), (4, 40_000, 15_000), (7, 33_000, 17_000), (12, 34_000, 16_000)])
def test_softmin_fwd_large_launch_paths(cuda, D, N, M):
    """M >= 65536: XCD-aware 1-D grid with 8-32 column splits.  N x M >= 5e8 pairs with >= 32768 rows: the 8-wavefront
    workgroups (2 row tiles per wavefront up to D = 8).  100 sampled rows against the float64 oracle."""
    x, y, h = _clouds(D + N, N, M, D)
    eps = 0.07**2
    rows = np.unique(np.r_[0, N - 1, np.random.default_rng(1).integers(0, N, 100)])
    ref = o64.softmin(eps, x, y, h, rows=rows, device=cuda)
    out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda)).cpu().numpy()
    assert np.isfinite(out).all()
    assert np.abs(out[rows] - ref).max() < _tol(ref, D)
    alt = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), flags=hip.FLAG_NO_SPLIT).cpu().numpy()
    assert np.abs(out - alt).max() < 2 * _tol(ref, D)             # every row, against the unsplit launch
    h2 = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), flags=hip.FLAG_F16X2).cpu().numpy()      # the f16 x 2 layout on the same paths
    assert np.abs(h2[rows] - ref).max() < _tol(ref, D) and np.abs(h2 - out).max() < 2 * _tol(ref, D)


@pytest.mark.parametrize("D,N,M", [(3, 33_000, 66_001), (2, 40_000, 100_003), (1, 70_000, 80_000)])
def test_mid_size_dense_launches_with_a_free_number_of_splits(cuda, D, N, M):
    """f16 x 2 layout, D <= 3, >= 32768 rows, 65536 <= M with all packed columns (32 bytes each) inside one XCD's L2 (<= 3.5 MB): the
    launch takes the split count that fills the chip's rounds of resident workgroups (glhip_mapreduce.h: free_splits — 23 splits
    for 65 row blocks of 512 rows here, 13 at N = M = 1e5), not a multiple of 8, on the plain 3-D grid with pre-packed columns.
    Soft-min and fused half-step: sampled rows against the float64 oracle, every row against the bf16 x 3 launch of the same call."""
    x, y, h = _clouds(11 * D + 5, N, M, D)
    h[-3:] += 25.0                                   # late maxima in the last, short split
    eps = 0.07**2
    rows = np.unique(np.r_[0, 511, 512, N - 1, np.random.default_rng(3).integers(0, N, 100)])
    xt, yt, ht = _t(x, cuda), _t(y, cuda), _t(h, cuda)
    ref = o64.softmin(eps, x, y, h, rows=rows, device=cuda)
    out = hip.softmin(eps, xt, yt, ht, flags=hip.FLAG_F16X2)
    assert np.abs(out.cpu().numpy()[rows] - ref).max() < _tol(ref, D)
    assert (out - hip.softmin(eps, xt, yt, ht)).abs().max().item() < 2 * _tol(ref, D)
    pot = _t((np.random.default_rng(4).standard_normal(M) * 0.05).astype(np.float32), cuda)
    prev = _t(np.random.default_rng(5).standard_normal(N).astype(np.float32), cuda)
    fused = hip.sinkhorn_step(eps, xt, yt, ht, pot, prev, 0.8, flags=hip.FLAG_F16X2)
    unfused = 0.5 * (prev + 0.8 * hip.softmin(eps, xt, yt, ht + pot / eps, flags=hip.FLAG_F16X2))
    assert (fused - unfused).abs().max().item() < 2e-6


@pytest.mark.parametrize("D,N,M", [(4, 33_000, 70_001), (8, 34_000, 66_000), (13, 33_333, 65_537), (16, 33_000, 70_001)])
def test_prepacked_columns_and_lds_dma_path(cuda, D, N, M):
    """Big dense launches (>= 5e8 pairs, >= 32768 rows, M >= 65536: 8-wavefront workgroups on the XCD-aware grid) run on PRE-PACKED
    columns: `xd_pack_kernel` writes the bf16x3 records of every column once and the reducing workgroups stream whole tiles into
    two LDS buffers with `global_load_lds_dwordx4` (glhip_softmin_xd.h).  M not a multiple of 32 (neutral padding columns, a short
    last tile in the last split), late maxima, the fused half-step and the gaussian product (weights travel by 4-byte LDS-DMA),
    against the float64 oracle on sampled rows — among them the first and last row blocks."""
    x, y, h = _clouds(3 * D + 1, N, M, D)
    h[-5:] += 30.0                                   # late maxima in the very last (short) tile
    eps = 0.07**2 * D / 3
    rows = np.unique(np.r_[0, 1, 255, 256, 511, 512, N - 513, N - 1, np.random.default_rng(2).integers(0, N, 120)])
    sel = torch.from_numpy(rows).to(cuda)
    xt, yt, ht = _t(x, cuda), _t(y, cuda), _t(h, cuda)
    ref = o64.softmin(eps, x, y, h, rows=rows, device=cuda)
    out = hip.softmin(eps, xt, yt, ht)
    assert np.abs(out[sel].cpu().numpy() - ref).max() < _tol(ref, D)
    ws = hip.load_library().glhip_workspace_bytes(1, N, M, D, 0)
    assert ws >= ((M + 31) // 32) * 32 * 2 * ((6 * (D + 1) + 15) // 16) * 16   # the packed records are part of the workspace contract
    # same bits from a second call (the pack kernel and the reducing kernel derive the launch centre independently)
    assert torch.equal(out, hip.softmin(eps, xt, yt, ht))
    pot = _t((np.random.default_rng(4).standard_normal(M) * 0.05).astype(np.float32), cuda)
    prev = _t(np.random.default_rng(5).standard_normal(N).astype(np.float32), cuda)
    fused = hip.sinkhorn_step(eps, xt, yt, ht, pot, prev, 0.8)
    unfused = 0.5 * (prev + 0.8 * hip.softmin(eps, xt, yt, ht + pot / eps))
    assert (fused - unfused).abs().max().item() < 2e-6
    v = _t(((np.random.default_rng(6).random(M) - 0.3) / M).astype(np.float32), cuda)      # signed weights
    blur = 0.2 * math.sqrt(D / 3)
    refk = o64.kconv("gaussian", x, y, v.cpu().numpy(), blur, rows=rows, device=cuda)
    outk = hip.kernel_conv("gaussian", xt, yt, v, blur)
    assert relerr(outk[sel].cpu().numpy(), refk) < 1e-4
    # ... and the same three launches on the f16 x 2 layout (half the record bytes: other tile sizes, other split counts)
    H2 = hip.FLAG_F16X2
    assert np.abs(hip.softmin(eps, xt, yt, ht, flags=H2)[sel].cpu().numpy() - ref).max() < _tol(ref, D)
    assert (hip.sinkhorn_step(eps, xt, yt, ht, pot, prev, 0.8, flags=H2) - unfused).abs().max().item() < 2 * _tol(ref, D)
    assert relerr(hip.kernel_conv("gaussian", xt, yt, v, blur, flags=H2)[sel].cpu().numpy(), refk) < 1e-4


def test_softmin_batched_bf16_and_fused_step(cuda):
    B, N, M, D = 3, 257, 300, 6
    x, y, logw = _clouds(3, N, M, D, B=B)
    xb, yb = _t(x, cuda).bfloat16(), _t(y, cuda).bfloat16()
    eps = 0.02
    ref = np.stack([oracle_c.softmin(eps, xb[b].float().cpu().numpy(), yb[b].float().cpu().numpy(), logw[b], 2) for b in range(B)])
    out = hip.softmin(eps, xb, yb, _t(logw, cuda)).cpu().numpy()
    assert out.shape == (B, N) and np.abs(out - ref).max() < _tol(ref, D)
    # glhip_sinkhorn_step == (prev + damping * softmin(eps, C, logw + pot / eps)) / 2
    rng = np.random.default_rng(8)
    pot = (rng.standard_normal(logw.shape) * 0.05).astype(np.float32)
    prev = rng.standard_normal(x.shape[:-1]).astype(np.float32)
    damping = 0.8
    xt, yt = _t(x, cuda), _t(y, cuda)
    unfused = 0.5 * (_t(prev, cuda) + damping * hip.softmin(eps, xt, yt, _t(logw + pot / np.float32(eps), cuda)))
    fused = hip.sinkhorn_step(eps, xt, yt, _t(logw, cuda), _t(pot, cuda), _t(prev, cuda), damping)
    assert (fused - unfused).abs().max().item() < 2e-6
    first = hip.sinkhorn_step(eps, xt, yt, _t(logw, cuda), None, None, damping).cpu().numpy()
    ref1 = damping * np.stack([oracle_c.softmin(eps, x[b], y[b], logw[b], 2) for b in range(B)])
    assert np.abs(first - ref1).max() < _tol(ref1, D)
    # nothing has a fused kernel beyond D = 16 (nor block-sparse p = 1 beyond D = 3), and NO_MFMA / DIRECT switch the matrix-core kernels
    # off: hip.sinkhorn_step then composes the soft-min kernel with torch arithmetic (round 3 raised NotImplementedError here);
    # dense p = 1 is one launch since round 5 (glhip_dist_xd.h) and must agree with the composition all the same
    assert hip.fused_step_applies(D, 1) and not hip.fused_step_applies(D, 1, 0, True)
    assert not hip.fused_step_applies(17, 2) and not hip.fused_step_applies(D, 2, hip.FLAG_NO_MFMA)
    for kw in (dict(p=1), dict(flags=hip.FLAG_NO_MFMA), dict(flags=hip.FLAG_DIRECT)):
        p = kw.get("p", 2)
        want = 0.5 * (_t(prev, cuda) + damping * hip.softmin(eps, xt, yt, _t(logw + pot / np.float32(eps), cuda), **kw))
        got = hip.sinkhorn_step(eps, xt, yt, _t(logw, cuda), _t(pot, cuda), _t(prev, cuda), damping, **kw)
        assert got.shape == want.shape and (got - want).abs().max().item() < 2e-6, kw
        if p == 2:
            assert (got - unfused).abs().max().item() < 3e-5 * max(1.0, unfused.abs().max().item()), kw
    x17, y17 = torch.rand(10, 17, device=cuda), torch.rand(12, 17, device=cuda)
    got = hip.sinkhorn_step(eps, x17, y17, torch.zeros(12, device=cuda), None, None, 0.5)
    assert torch.allclose(got, 0.5 * hip.softmin(eps, x17, y17, torch.zeros(12, device=cuda)), rtol=0, atol=1e-6)


@pytest.mark.parametrize("D", [4, 5, 16])
def test_softmin_lazy_max_and_infinities(cuda, D):
    """Late maxima, -inf / -1e5 dual values, exponents that climb faster than the speculative tile pass tolerates."""
    N, M = 130, 2500
    x, y, h = _clouds(9 + D, N, M, D)
    h[:] = -50.0
    h[-1] = 80.0
    h[5] = -np.inf
    h[6] = -100000.0
    eps = 0.05**2
    ref = oracle_c.softmin(eps, x, y, h, 2)
    for flags in (0, hip.FLAG_NO_SPLIT, hip.FLAG_F16X2, hip.FLAG_F16X2 | hip.FLAG_NO_SPLIT):
        out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), flags=flags).cpu().numpy()
        assert np.isfinite(out).all() and np.abs(out - ref).max() < _tol(ref, D), flags
    h2 = (np.arange(M) // 64 * 48.0).astype(np.float32)
    h2[M // 2:] -= 3000.0
    h2[-3] = 5000.0
    ref2 = oracle_c.softmin(eps, x, y, h2, 2)
    for flags in (0, hip.FLAG_F16X2):
        out2 = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h2, cuda), flags=flags).cpu().numpy()
        assert np.isfinite(out2).all() and relerr(out2, ref2) < 2e-6, flags
    # a measure without any mass: the reference returns +inf; every kernel of the library (D <= 3 included) returns a huge finite
    # potential instead — the neutral padding columns of the last tile carry -1e30, not -inf, and are all that is left
    allinf = hip.softmin(eps, _t(x, cuda), _t(y, cuda), torch.full((M,), -math.inf, device=cuda)).cpu().numpy()
    assert (allinf > 1e20).all()
    # (the f16 x 2 layout floors every exponent at -5e5 and recognises a row that never left the floor: +inf, like the reference)
    allinf = hip.softmin(eps, _t(x, cuda), _t(y, cuda), torch.full((M,), -math.inf, device=cuda), flags=hip.FLAG_F16X2).cpu().numpy()
    assert np.isposinf(allinf).all()


def _random_ranges(rng, N, M, ci, cj, density, dev):
    cut_i = np.sort(rng.choice(np.arange(1, N), ci - 1, replace=False))
    cut_j = np.sort(rng.choice(np.arange(1, M), cj - 1, replace=False))
    ri = np.stack([np.r_[0, cut_i], np.r_[cut_i, N]], 1).astype(np.int32)
    rj = np.stack([np.r_[0, cut_j], np.r_[cut_j, M]], 1).astype(np.int32)
    keep = rng.random((ci, cj)) < density
    keep[0, :] = False      # one row block with nothing to reduce over
    keep[1, :] = True
    rg = from_matrix(torch.from_numpy(ri).to(dev), torch.from_numpy(rj).to(dev), torch.from_numpy(keep).to(dev))
    tup = tuple(t.cpu().numpy() for t in (rg.ranges_i, rg.slices_i, rg.redranges_j))
    return rg, tup, ri


@pytest.mark.parametrize("D,ci,cj", [(4, 9, 11), (9, 9, 11), (4, 60, 70), (9, 60, 70), (16, 60, 70)])
def test_block_sparse_softmin_and_gaussian(cuda, D, ci, cj):
    # 60 x 70 clusters: row clusters of ~40 points, whose tiles gather several column intervals (SplitInfo::gather)
    rng = np.random.default_rng(17)
    N, M = 2300, 2600
    x, y, h = _clouds(23, N, M, D)
    rg, tup, ri = _random_ranges(rng, N, M, ci, cj, 0.4, cuda)
    eps = 0.02
    ref = oracle_c.softmin(eps, x, y, h, 2, ranges=tup)
    empty = slice(ri[0, 0], ri[0, 1])
    live = np.ones(N, bool)
    live[empty] = False
    for flags in (0, hip.FLAG_NO_SPLIT, hip.FLAG_F16X2, hip.FLAG_F16X2 | hip.FLAG_NO_SPLIT):
        out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), ranges=rg, flags=flags).cpu().numpy()
        assert np.isposinf(out[empty]).all() and np.isposinf(ref[empty]).all(), flags
        assert np.abs(out[live] - ref[live]).max() < _tol(ref[live], D), flags
    v = (np.abs(h) / M).astype(np.float32)
    blur = 0.3
    refk = oracle_c.kconv("gaussian", x, y, v, blur, ranges=tup)
    for flags in (0, hip.FLAG_F16X2):
        k = hip.kernel_conv("gaussian", _t(x, cuda), _t(y, cuda), _t(v, cuda), blur, ranges=rg, flags=flags).cpu().numpy()
        assert (k[empty] == 0).all() and relerr(k, refk) < 1e-4, flags


@pytest.mark.parametrize("D", [4, 6, 11, 16])
@pytest.mark.parametrize("N,M,B", [(300, 257, None), (1030, 2100, None), (150, 170, 3), (700, 70_001, None)])
def test_gaussian_product_vs_oracle(cuda, D, N, M, B):
    x, y, v = _clouds(77 + D, N, M, D, B=B)
    v = np.abs(v) / M
    v[..., ::7] *= -1.0                                    # signed weights are legal
    blur = 0.25 * math.sqrt(D / 3)
    one = lambda xa, ya, va: oracle_c.kconv("gaussian", xa, ya, va, blur)      # noqa: E731
    ref = one(x, y, v) if B is None else np.stack([one(x[b], y[b], v[b]) for b in range(B)])
    bound = one(x, y, np.abs(v)) if B is None else np.stack([one(x[b], y[b], np.abs(v[b])) for b in range(B)])
    tol = 3e-6 * np.abs(ref).max() + 2.4e-7 * D / blur**2 * np.abs(bound).max()     # as in test_hip_kernels.py
    for flags in (0, hip.FLAG_NO_SPLIT, hip.FLAG_F16X2, hip.FLAG_NO_MFMA):
        out = hip.kernel_conv("gaussian", _t(x, cuda), _t(y, cuda), _t(v, cuda), blur, flags=flags).cpu().numpy()
        assert np.abs(out - ref).max() < tol, flags


@pytest.mark.parametrize("name", ["sinkhorn_p2_d5", "gaussian_d6"])
def test_reference_goldens_in_higher_dimension(cuda, name):
    """The two reference-generated golden cases with D > 3 (tests/golden/make_golden.py), through SamplesLoss(backend="online"):
    loss <= 1e-4 of the reference's float64 value, with the forward reductions on the matrix cores."""
    rec = load_golden(name)
    x, y = _t(rec["x"].astype(np.float32), cuda), _t(rec["y"].astype(np.float32), cuda)
    a, b = _t(rec["a"].astype(np.float32), cuda), _t(rec["b"].astype(np.float32), cuda)
    assert x.shape[-1] > 3
    xg = x.clone().requires_grad_(True)
    L = SamplesLoss(backend="online", **rec["kwargs"])(a, xg, b, y)
    (gx,) = torch.autograd.grad(L, [xg])
    assert abs(L.item() - float(rec["loss_f64"])) <= 1e-4 * abs(float(rec["loss_f64"]))
    assert relerr(gx.cpu().numpy(), rec["gx_f64"]) < 1e-4


def test_multiscale_4d_with_user_labels(cuda):
    """The reference's recipe for D > 3 (plot_optimal_transport_cluster.py:155-166): clusters given as labels — here voxels of
    the three spatial coordinates of (position, feature) points — and the two-scale solver on the 4-D clouds.  Against the online
    backend on the same clouds (the truncation error of a two-scale run at truncate = 5 is ~1e-3 of the loss) and, for the
    block-sparse reductions it runs, see test_block_sparse_softmin_and_gaussian."""
    g = torch.Generator().manual_seed(7)
    N, M = 6000, 7000
    x = torch.rand(N, 4, generator=g).to(cuda)
    y = (torch.rand(M, 4, generator=g) * torch.tensor([0.7, 0.7, 0.7, 1.0]) + torch.tensor([0.2, 0.2, 0.2, 0.0])).to(cuda)
    def lab(t):       # compact labels 0..C-1 in voxel order, as pykeops' grid_cluster would give for the spatial coordinates
        code = ((t[:, :3] / 0.2).floor().long() * torch.tensor([36, 6, 1], device=cuda)).sum(1)
        return torch.unique(code, return_inverse=True)[1].int()

    kw = dict(p=2, blur=0.05, scaling=0.7)
    xg = x.clone().requires_grad_(True)
    a, b = torch.full((N,), 1.0 / N, device=cuda), torch.full((M,), 1.0 / M, device=cuda)
    Lm = SamplesLoss("sinkhorn", backend="multiscale", **kw)(lab(x), a, xg, lab(y), b, y)
    (gm,) = torch.autograd.grad(Lm, [xg])
    xo = x.clone().requires_grad_(True)
    Lo = SamplesLoss("sinkhorn", backend="online", **kw)(xo, y)
    (go,) = torch.autograd.grad(Lo, [xo])
    assert abs(Lm.item() - Lo.item()) < 5e-3 * abs(Lo.item())
    # the two schemes stop at differently converged potentials (coarse start + extrapolation vs a full annealing): their gradient
    # fields agree in direction and size, not digit for digit
    assert torch.isfinite(gm).all() and (gm - go).norm() < 0.5 * go.norm()


# ---- gradients on the transposed 32x32x16 kernel (csrc/glhip_wsum_t32.h): 4 <= D <= 16 -----

# (dimension, flags): the default bf16 x 3 layout, and the f16 x 2 one — on which D >= 7 soft-min gradients (and value + gradient) run
# wsum_t32q_kernel, their weighted sums on the matrix cores too (round 5)
T32_CASES = [(4, 0), (5, 0), (8, 0), (9, 0), (16, 0), (5, hip.FLAG_F16X2), (7, hip.FLAG_F16X2), (8, hip.FLAG_F16X2), (12, hip.FLAG_F16X2),
             (16, hip.FLAG_F16X2)]


@pytest.mark.parametrize("D,flags", T32_CASES)
@pytest.mark.parametrize("N,M,B", [(300, 257, None), (1030, 2100, None), (257, 300, 3), (700, 70_001, None)])
def test_softmin_gradient_transposed_kernel(cuda, D, flags, N, M, B):
    if M > 50_000 and D not in (4, 16):
        pytest.skip("the many-column launch is exercised for two dimensions")
    x, y, h = _clouds(N + D, N, M, D, B=B)
    g = np.random.default_rng(6).standard_normal(x.shape[:-1]).astype(np.float32)
    eps = 0.1 * D / 3 if M < 50_000 else 0.05**2 * D
    one = lambda xa, ya, ha, ga: oracle_c.softmin_grad_x(eps, xa, ya, ha, ga, 2)        # noqa: E731
    ref = one(x, y, h, g) if B is None else np.stack([one(x[b], y[b], h[b], g[b]) for b in range(B)])
    # Overlapping clouds at a small temperature: the gradient is a small difference x_i - sum_j P_ij y_j (|g| ~ 0.1), and each
    # plan weight carries the ~1e-5 error of the expanded exponent: a few 1e-5 of the largest entry (measured the same for the
    # 16x16x32 kernel of D <= 3 on such inputs).
    tol = 2e-5 if M < 50_000 else 1e-4
    errs = {}
    for fl in (flags, flags | hip.FLAG_NO_SPLIT):
        xt = _t(x, cuda).requires_grad_(True)
        out = hip.softmin(eps, xt, _t(y, cuda), _t(h, cuda), flags=fl)
        (gx,) = torch.autograd.grad(out, [xt], grad_outputs=_t(g, cuda))
        errs[fl] = relerr(gx.cpu().numpy(), ref)
        assert errs[fl] < tol, (fl, errs)


@pytest.mark.parametrize("D,flags", T32_CASES)
def test_softmin_value_and_gradient_transposed_kernel(cuda, D, flags):
    """glhip_softmin_fwd_grad on the transposed kernel: exact value whatever the guess (within the margin), unit gradient."""
    N, M = 700, 900
    x, y, h = _clouds(40 + D, N, M, D)
    eps = 0.05 * D / 3
    ref = oracle_c.softmin(eps, x, y, h, 2)
    refg = oracle_c.softmin_grad_x(eps, x, y, h, np.ones(N, np.float32), 2)
    rng = np.random.default_rng(2)
    for margin in (1e-3 * eps, 3 * eps):
        guess = (ref + margin * (2 * rng.random(N) - 1)).astype(np.float32)
        out, unit = hip.softmin_fwd_grad_raw(_t(x, cuda)[None].contiguous(), _t(y, cuda)[None].contiguous(), _t(h, cuda)[None].contiguous(),
                                             _t(guess, cuda)[None].contiguous(), 1.0001 * margin + 1e-7, eps, flags=flags)
        assert np.abs(out[0].cpu().numpy() - ref).max() < _tol(ref, D) + 4e-7 * margin
        assert relerr(unit[0].cpu().numpy(), refg) < 2e-5


@pytest.mark.parametrize("D,flags", T32_CASES)
@pytest.mark.parametrize("N,M,B", [(310, 270, None), (257, 300, 3), (1030, 70_001, None)])
def test_gaussian_gradient_and_one_pass_transposed_kernel(cuda, D, flags, N, M, B):
    if M > 50_000 and D not in (4, 16):
        pytest.skip("the many-column launch is exercised for three dimensions")
    x, y, v = _clouds(90 + D, N, M, D, B=B)
    v = (np.abs(v) / M).astype(np.float32)
    v[..., ::5] *= -1.0
    g = np.random.default_rng(3).standard_normal(x.shape[:-1]).astype(np.float32)
    blur = 0.25 * math.sqrt(D / 3)
    per = lambda f, *a: f(*a) if B is None else np.stack([f(*[t[b] for t in a]) for b in range(B)])      # noqa: E731
    ref = per(lambda xa, ya, va: oracle_c.kconv("gaussian", xa, ya, va, blur), x, y, v)
    refg = per(lambda xa, ya, va, ga: oracle_c.kconv_grad_x("gaussian", xa, ya, va, ga, blur), x, y, v, g)
    xb, yb, vb = (_t(t, cuda) if B is not None else _t(t, cuda)[None].contiguous() for t in (x, y, v))
    gb = _t(g, cuda) if B is not None else _t(g, cuda)[None].contiguous()
    shp = ref.shape
    for fl in (flags, flags | hip.FLAG_NO_SPLIT):
        gx = hip.kernel_conv_bwd_x_raw(hip.GAUSSIAN, xb, yb, vb, gb, blur, flags=fl).reshape(shp + (D,)).cpu().numpy()
        assert relerr(gx, refg) < 1e-4, fl
        out, unit = hip.kernel_conv_fwd_grad_raw(hip.GAUSSIAN, xb, yb, vb, blur, flags=fl)
        assert relerr(out.reshape(shp).cpu().numpy(), ref) < 1e-4, fl
        assert relerr((gb.unsqueeze(-1) * unit).reshape(shp + (D,)).cpu().numpy(), refg) < 1e-4, fl


@pytest.mark.parametrize("D,flags", [(4, 0), (9, 0), (5, hip.FLAG_F16X2), (9, hip.FLAG_F16X2), (16, hip.FLAG_F16X2)])
def test_block_sparse_gradients_transposed_kernel(cuda, D, flags):
    rng = np.random.default_rng(19)
    N, M = 2300, 2600
    x, y, h = _clouds(27, N, M, D)
    rg, tup, ri = _random_ranges(rng, N, M, 9, 11, 0.4, cuda)
    live = np.ones(N, bool)
    live[ri[0, 0]:ri[0, 1]] = False
    g = rng.standard_normal(N).astype(np.float32)
    g[~live] = 0
    eps = 0.02 * D / 3
    xt = _t(x, cuda).requires_grad_(True)
    out = hip.softmin(eps, xt, _t(y, cuda), _t(h, cuda), ranges=rg, flags=flags)
    (gx,) = torch.autograd.grad(out[torch.from_numpy(live).to(cuda)], [xt], grad_outputs=_t(g[live], cuda))
    ref = oracle_c.softmin_grad_x(eps, x, y, h, g, 2, ranges=tup)
    assert relerr(gx.cpu().numpy()[live], ref[live]) < 2e-5
    v = (np.abs(h) / M).astype(np.float32)
    blur = 0.3 * math.sqrt(D / 3)
    xg = _t(x, cuda).requires_grad_(True)
    k = hip.kernel_conv("gaussian", xg, _t(y, cuda), _t(v, cuda), blur, ranges=rg, flags=flags)      # fused product + gradient
    (gk,) = torch.autograd.grad(k, [xg], grad_outputs=_t(g, cuda))
    assert relerr(k.detach().cpu().numpy(), oracle_c.kconv("gaussian", x, y, v, blur, ranges=tup)) < 1e-4
    assert relerr(gk.cpu().numpy(), oracle_c.kconv_grad_x("gaussian", x, y, v, g, blur, ranges=tup)) < 1e-4


# ---- round 5: the distance reductions of 4 <= D <= 16 on the matrix cores (csrc/glhip_dist_xd.h) ---------------------------------

@pytest.mark.parametrize("D", XD)
@pytest.mark.parametrize("N,M,B", [(300, 257, None), (1030, 2100, None), (1, 1, None), (5, 3000, None), (260, 300, 3)])
def test_distance_reductions_xd_vs_oracle(cuda, D, N, M, B):
    """Soft-min p = 1 (plain and fused half-step), laplacian and energy products: squared distances from the MFMA chain, with coincident
    and near pairs (the exact path; the debiasing terms of a loss are x against x), -inf dual values, batches, column splits or not —
    against the float64 C oracle, and the one-thread-per-row kernel (GLHIP_FLAG_NO_MFMA) as a second opinion."""
    x, y, h = _clouds(7 * D + N + M, N, M, D, B)
    k = min(N, M) // 3
    y[..., :k, :] = x[..., :k, :]                       # coincident pairs: the floor of utils.py:61
    y[..., k:2 * k, :] = x[..., k:2 * k, :] + 3e-5       # near pairs
    if M > 11:
        h[..., ::11] = -np.inf
    v = (np.random.default_rng(1).random(h.shape) / M).astype(np.float32)
    eps, blur = 0.05, 0.2
    bs = range(B) if B is not None else [None]
    sel = (lambda a, b: a if b is None else a[b])
    ref_f = np.stack([oracle_c.softmin(eps, sel(x, b), sel(y, b), sel(h, b), 1) for b in bs]).reshape(h.shape[:-1] + (N,))
    ref_l = np.stack([oracle_c.kconv("laplacian", sel(x, b), sel(y, b), sel(v, b), blur) for b in bs]).reshape(ref_f.shape)
    ref_e = np.stack([oracle_c.kconv("energy", sel(x, b), sel(y, b), sel(v, b), blur) for b in bs]).reshape(ref_f.shape)
    xt, yt, ht, vt = (_t(a, cuda) for a in (x, y, h, v))
    for flags in (0, hip.FLAG_NO_SPLIT, hip.FLAG_NO_MFMA):
        tol = 1 if flags != hip.FLAG_NO_MFMA else 12      # the generic kernel sums |x - y|^2 coordinate by coordinate in fp32
        f = hip.softmin(eps, xt, yt, ht, p=1, flags=flags).cpu().numpy()
        assert np.abs(f - ref_f).max() < tol * 2e-6 * max(1.0, np.abs(ref_f).max()), flags
        assert relerr(hip.kernel_conv("laplacian", xt, yt, vt, blur, flags=flags).cpu().numpy(), ref_l) < tol * 5e-6, flags
        assert relerr(hip.kernel_conv("energy", xt, yt, vt, blur, flags=flags).cpu().numpy(), ref_e) < tol * 5e-6, flags
    # gradients with respect to the rows (dist_xd_grad_kernel; GLHIP_FLAG_NO_MFMA: the one-thread-per-row kernel), zero direction at
    # coincident pairs like autograd through sqrt(clamp_min(d2, 1e-8))
    g = np.random.default_rng(5).standard_normal(ref_f.shape).astype(np.float32)
    hf = np.where(np.isinf(h), -40.0, h).astype(np.float32)          # finite dual values: every row keeps some mass
    rg_f = np.stack([oracle_c.softmin_grad_x(eps, sel(x, b), sel(y, b), sel(hf, b), sel(g, b), 1) for b in bs]).reshape(x.shape)
    rg_l = np.stack([oracle_c.kconv_grad_x("laplacian", sel(x, b), sel(y, b), sel(v, b), sel(g, b), blur) for b in bs]).reshape(x.shape)
    rg_e = np.stack([oracle_c.kconv_grad_x("energy", sel(x, b), sel(y, b), sel(v, b), sel(g, b), blur) for b in bs]).reshape(x.shape)
    for flags in (0, hip.FLAG_NO_SPLIT, hip.FLAG_NO_MFMA):
        xg = xt.clone().requires_grad_(True)
        (gx,) = torch.autograd.grad(hip.softmin(eps, xg, yt, _t(hf, cuda), p=1, flags=flags), [xg], grad_outputs=_t(g, cuda))
        assert relerr(gx.cpu().numpy(), rg_f) < 3e-5, flags
        for kind, want in (("laplacian", rg_l), ("energy", rg_e)):
            xg = xt.clone().requires_grad_(True)
            (gx,) = torch.autograd.grad(hip.kernel_conv(kind, xg, yt, vt, blur, flags=flags), [xg], grad_outputs=_t(g, cuda))
            assert relerr(gx.cpu().numpy(), want) < 3e-5, (kind, flags)
    # fused half-step (dense p = 1: one launch)
    rng = np.random.default_rng(3)
    pot, prev = rng.standard_normal(h.shape).astype(np.float32) * 0.05, rng.standard_normal(ref_f.shape).astype(np.float32)
    logw = np.where(np.isinf(h), -30.0, h).astype(np.float32)
    got = hip.sinkhorn_step(eps, xt, yt, _t(logw, cuda), _t(pot, cuda), _t(prev, cuda), 0.8, p=1).cpu().numpy()
    want = np.stack([oracle_c.softmin(eps, sel(x, b), sel(y, b), sel(logw, b) + sel(pot, b) / eps, 1) for b in bs]).reshape(ref_f.shape)
    assert np.abs(got - 0.5 * (prev + 0.8 * want)).max() < 2e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("D", [4, 5, 9])
@pytest.mark.parametrize("ratio", [0.005, 0.002])
def test_distance_reductions_xd_small_blur(cuda, D, ratio):
    """blur / diameter = 0.005 and 0.002 (round-5 advice): the squared distance of the MFMA chain carries ~2^-23 t^2 R^2, so the
    EXPONENT of a pair just above the near-pair guard (d = R / 16) is off by ~2^-20 t R — it grows like diameter / blur — but a
    potential is eps times the logarithm: its error stays ~2^-20 log2(e) R whatever the blur (the bound written in include/glhip.h).
    Soft-min p = 1 on x against y and on x against itself (the debiasing term: one coincident pair per row, neighbours at every
    distance), laplacian product: sampled rows against the chunked float64 oracle."""
    N, M = 6000, 9000
    x, y, h = _clouds(31 * D, N, M, D)
    diam = math.sqrt(D)
    eps = ratio * diam                                   # p = 1: eps = blur
    h = (h * 0.1).astype(np.float32)
    rows = np.unique(np.r_[0, N - 1, np.random.default_rng(2).integers(0, N, 150)])
    bound = 2.0 ** -20 * 1.4427 * diam                   # the documented bound on a potential, 4x margin below
    xt, yt = _t(x, cuda), _t(y, cuda)
    f = hip.softmin(eps, xt, yt, _t(h, cuda), p=1).cpu().numpy()
    assert np.abs(f[rows] - o64.softmin(eps, x, y, h, p=1, rows=rows, device=cuda)).max() < 4 * bound
    hs = np.ascontiguousarray(h[:N])
    f = hip.softmin(eps, xt, xt, _t(hs, cuda), p=1).cpu().numpy()
    assert np.abs(f[rows] - o64.softmin(eps, x, x, hs, p=1, rows=rows, device=cuda)).max() < 4 * bound
    # kernel products: the error of a kernel VALUE is relative (2^-20 t R at the guard, random sign); sums of 9000 of them
    v = (np.random.default_rng(3).random(M) / M).astype(np.float32)
    blur = 0.02 * diam                                   # (at ratio 0.002 a laplacian product of 9000 points is one or two terms)
    k = hip.kernel_conv("laplacian", xt, yt, _t(v, cuda), blur).cpu().numpy()
    assert relerr(k[rows], o64.kconv("laplacian", x, y, v, blur, rows=rows, device=cuda)) < 2e-5


def test_distance_reductions_xd_many_columns_and_self_term(cuda):
    """70 001 columns (column splits + merge) and a self-term (x against x: every row has one coincident pair), D = 6, 100 sampled
    rows against the chunked float64 oracle."""
    D, N, M = 6, 9000, 70_001
    x, y, h = _clouds(5, N, M, D)
    rows = np.unique(np.r_[0, N - 1, np.random.default_rng(1).integers(0, N, 100)])
    eps = 0.1
    out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), p=1).cpu().numpy()
    assert np.abs(out[rows] - o64.softmin(eps, x, y, h, p=1, rows=rows, device=cuda)).max() < 2e-6
    hs = h[:N]
    out = hip.softmin(eps, _t(x, cuda), _t(x, cuda), _t(hs, cuda), p=1).cpu().numpy()
    assert np.abs(out[rows] - o64.softmin(eps, x, x, hs, p=1, rows=rows, device=cuda)).max() < 2e-6
    v = (np.random.default_rng(2).random(N) / N).astype(np.float32)
    e = hip.kernel_conv("energy", _t(x, cuda), _t(x, cuda), _t(v, cuda), 0.1).cpu().numpy()
    assert relerr(e[rows], o64.kconv("energy", x, x, v, 0.1, rows=rows, device=cuda)) < 5e-6
