"""GPU parity of every C-ABI entry point against the CPU oracle (float64 restatement of the reference).

Tolerances: the kernels compute in fp32, the oracle in fp64.
* DIRECT form (p=1, or p=2 with FLAG_DIRECT): max-norm relative error < 2e-6.
* EXPANDED form (p=2 default): the exponent is a sum of terms of size diam^2/eps, so the potential
  carries an ABSOLUTE error of a few 2^-24 * diam^2 whatever eps is (the reference's dense fp32 cost
  |x|^2 - 2x.y + |y|^2 has the same property).  Bound used: 4e-7 * diam^2 plus the 2e-6 relative term.
Both are far inside the 1e-4 relative budget BASELINE.json states for the loss; the loss-level check is
tests/test_samples_loss_gpu.py.
"""

import ctypes

import numpy as np
import pytest
import torch

from conftest import relerr
from geomloss_amd import hip
from geomloss_amd.cluster import from_matrix
from oracle import oracle_c, oracle_np

pytestmark = pytest.mark.gpu


def _clouds(seed, N, M, D, B=None, offset=0.0):
    rng = np.random.default_rng(seed)
    shp = (lambda n: (n, D)) if B is None else (lambda n: (B, n, D))
    x = rng.random(shp(N)).astype(np.float32) + offset
    y = (rng.random(shp(M)) * 0.8 + 0.1).astype(np.float32) + offset
    h = rng.standard_normal(shp(M)[:-1]).astype(np.float32)
    return x, y, h


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


SOFTMIN_SHAPES = [(300, 257, 3), (1030, 2100, 3), (513, 1025, 2), (700, 900, 1), (64, 8, 3), (1, 1, 3), (5, 3000, 2)]


@pytest.mark.parametrize("N,M,D", SOFTMIN_SHAPES)
@pytest.mark.parametrize("p", [2, 1])
@pytest.mark.parametrize("eps", [1.0, 0.05**2])
def test_softmin_fwd_vs_oracle(cuda, N, M, D, p, eps):
    x, y, h = _clouds(N + M + D, N, M, D)
    ref = oracle_c.softmin(eps, x, y, h, p)
    # every code path of the forward kernel: matrix-core / VALU exponents, with / without column splits
    for flags in (0, hip.FLAG_F32_MFMA, hip.FLAG_XDL16, hip.FLAG_NO_MFMA, hip.FLAG_NO_SPLIT, hip.FLAG_F32_MFMA | hip.FLAG_NO_SPLIT,
                  hip.FLAG_XDL16 | hip.FLAG_NO_SPLIT, hip.FLAG_NO_MFMA | hip.FLAG_NO_SPLIT, hip.FLAG_PREPACK,
                  hip.FLAG_PREPACK | hip.FLAG_NO_SPLIT, hip.FLAG_F16X2, hip.FLAG_F16X2 | hip.FLAG_NO_SPLIT, hip.FLAG_F16X2 | hip.FLAG_PREPACK):
        # (GLHIP_FLAG_F16X2: the two-piece f16 layout of the p = 2 exponents — one MFMA per block; in range here; ignored for p = 1)
        out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), p=p, flags=flags).cpu().numpy()
        assert np.abs(out - ref).max() < 4e-7 * D + 2e-6 * np.abs(ref).max(), flags  # diam^2 <= D on the unit cube
    out_d = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), p=p, flags=hip.FLAG_DIRECT).cpu().numpy()
    assert relerr(out_d, ref) < 2e-6


@pytest.mark.parametrize("N,M,D", [(700, 70001, 3), (300, 66000, 2)])
def test_softmin_fwd_many_columns(cuda, N, M, D):
    """M >= 65536 selects the 8-split XCD-aware grid and, for the default kernel, the pre-packed column records."""
    eps = 0.05**2
    x, y, h = _clouds(N + D, N, M, D)
    ref = oracle_c.softmin(eps, x, y, h, 2)
    for flags in (0, hip.FLAG_XDL16, hip.FLAG_NO_SPLIT, hip.FLAG_PREPACK, hip.FLAG_F16X2, hip.FLAG_F16X2 | hip.FLAG_PREPACK):
        out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), p=2, flags=flags).cpu().numpy()
        assert np.abs(out - ref).max() < 4e-7 * D + 2e-6 * np.abs(ref).max(), flags
    # sorted clouds, batched, through the fused half-step entry point
    xb, yb, hb = _clouds(5, 130, M, D, B=2)
    lw = np.full((2, M), -np.log(M), np.float32)
    ref_b = np.stack([oracle_c.softmin(eps, xb[k], yb[k], lw[k] + hb[k] / eps, 2) for k in range(2)])
    for flags in (0, hip.FLAG_PREPACK):
        out_b = hip.sinkhorn_step(eps, _t(xb, cuda), _t(yb, cuda), _t(lw, cuda), _t(hb, cuda), None, 1.0, flags=flags).cpu().numpy()
        assert np.abs(out_b - ref_b).max() < 4e-7 * D + 2e-6 * np.abs(ref_b).max(), flags


@pytest.mark.parametrize("N,M,D", [(300, 257, 3), (1030, 1100, 2), (200, 300, 1)])
@pytest.mark.parametrize("p", [2, 1])
def test_softmin_bwd_vs_oracle(cuda, N, M, D, p):
    eps = 0.01
    x, y, h = _clouds(7 * N + D, N, M, D)
    g = np.random.default_rng(3).standard_normal(N).astype(np.float32)
    ref = oracle_c.softmin_grad_x(eps, x, y, h, g, p)
    for flags in (0, hip.FLAG_NO_MFMA, hip.FLAG_NO_SPLIT, hip.FLAG_DIRECT):
        xt = _t(x, cuda).requires_grad_(True)
        out = hip.softmin(eps, xt, _t(y, cuda), _t(h, cuda), p=p, flags=flags)
        (gx,) = torch.autograd.grad(out, [xt], grad_outputs=_t(g, cuda))
        # fp32 weights 2^(u - lse): exponent error ~1e-5 at eps = 0.01 in the expanded forms
        assert relerr(gx.cpu().numpy(), ref) < 2e-5, flags


def test_softmin_batched_and_bf16(cuda):
    B, N, M, D = 5, 300, 400, 3
    x, y, h = _clouds(11, N, M, D, B=B)
    eps = 0.05**2
    xb, yb = _t(x, cuda).bfloat16(), _t(y, cuda).bfloat16()
    xr, yr = xb.float().cpu().numpy(), yb.float().cpu().numpy()
    for flags in (0, hip.FLAG_PREPACK, hip.FLAG_XDL16):   # per-workgroup packing / pre-packed records (one centre per item)
        out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), flags=flags).cpu().numpy()
        for b in range(B):
            assert np.abs(out[b] - oracle_c.softmin(eps, x[b], y[b], h[b], 2)).max() < 1.2e-6, flags
        # bf16 points: kernel widens to fp32, so parity is against the oracle on the bf16-rounded points
        out16 = hip.softmin(eps, xb, yb, _t(h, cuda), flags=flags).cpu().numpy()
        for b in range(B):
            assert np.abs(out16[b] - oracle_c.softmin(eps, xr[b], yr[b], h[b], 2)).max() < 1.2e-6, flags


def test_softmin_translation_robust(cuda):
    """Clouds far from the origin: the per-workgroup re-centring keeps the expanded form accurate."""
    x, y, h = _clouds(5, 600, 700, 3, offset=1000.0)
    eps = 0.05**2
    ref = oracle_c.softmin(eps, x, y, h, 2)
    out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda)).cpu().numpy()
    assert np.abs(out - ref).max() < 1e-5   # potentials are O(1): absolute tolerance


def test_softmin_rescale_branch_and_infinities(cuda):
    """Running max must survive a late, much larger exponent and -inf / -1e5 dual values."""
    N, M, D = 130, 2500, 3
    x, y, h = _clouds(9, N, M, D)
    h[:] = -50.0
    h[-1] = 80.0            # the maximum arrives in the last chunk of the last tile
    h[5] = -np.inf
    h[6] = -100000.0
    eps = 0.05**2
    ref = oracle_c.softmin(eps, x, y, h, 2)
    for flags in (0, 8, hip.FLAG_NO_MFMA, hip.FLAG_NO_SPLIT):
        out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), flags=flags).cpu().numpy()
        assert np.isfinite(out).all() and np.abs(out - ref).max() < 1.2e-6 + 2e-6 * np.abs(ref).max()
    # lazy-max stress for the matrix-core path: exponents climbing by ~70 (base 2) every 64 columns (so every
    # LDS tile overflows its speculative pass and is redone exactly), then a cliff, then one late spike
    h2 = (np.arange(M) // 64 * 48.0).astype(np.float32)
    h2[M // 2:] -= 3000.0
    h2[-3] = 5000.0
    ref2 = oracle_c.softmin(eps, x, y, h2, 2)
    for flags in (0, 8, hip.FLAG_NO_MFMA):
        out2 = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h2, cuda), flags=flags).cpu().numpy()
        assert np.isfinite(out2).all() and relerr(out2, ref2) < 2e-6


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
    tup_t = tuple(t.cpu().numpy() for t in (rg.ranges_j, rg.slices_j, rg.redranges_i))
    return rg, tup, tup_t, keep, ri


@pytest.mark.parametrize("p", [2, 1])
def test_softmin_block_sparse_vs_oracle(cuda, p):
    rng = np.random.default_rng(17)
    N, M, D = 2300, 2600, 3
    x, y, h = _clouds(23, N, M, D)
    rg, tup, _, keep, ri = _random_ranges(rng, N, M, 9, 11, 0.4, cuda)
    eps = 0.02
    ref = oracle_c.softmin(eps, x, y, h, p, ranges=tup)
    empty = slice(ri[0, 0], ri[0, 1])
    live = np.ones(N, bool)
    live[empty] = False
    for flags in (0, 8, hip.FLAG_NO_MFMA, hip.FLAG_NO_SPLIT, hip.FLAG_NO_MFMA | hip.FLAG_NO_SPLIT, hip.FLAG_PREPACK, hip.FLAG_XDL16,
                  hip.FLAG_F16X2, hip.FLAG_F16X2 | hip.FLAG_PREPACK, hip.FLAG_F16X2 | hip.FLAG_NO_SPLIT):
        out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), p=p, ranges=rg, flags=flags).cpu().numpy()
        assert np.isposinf(out[empty]).all() and np.isposinf(ref[empty]).all()   # LSE over the empty set
        assert np.abs(out[live] - ref[live]).max() < 1.2e-6 + 2e-6 * np.abs(ref[live]).max()
    g = rng.standard_normal(N).astype(np.float32)
    g[empty] = 0
    xt = _t(x, cuda).requires_grad_(True)
    o = hip.softmin(eps, xt, _t(y, cuda), _t(h, cuda), p=p, ranges=rg)
    (gx,) = torch.autograd.grad(o[torch.from_numpy(live).to(cuda)], [xt], grad_outputs=_t(g[live], cuda))
    refg = oracle_c.softmin_grad_x(eps, x, y, h, g, p, ranges=tup)
    assert relerr(gx.cpu().numpy()[live], refg[live]) < 5e-6


@pytest.mark.parametrize("D", [1, 2, 3])
@pytest.mark.parametrize("half", [False, True])
def test_softmin_block_sparse_small_row_blocks(cuda, D, half):
    """Row blocks of ~30 points (the reference's ~2000 clusters on clouds of up to 1e5 points): the forward kernel runs as 2-wavefront
    workgroups over gathered 256-column tiles (glhip_launch.h: launch_softmin_mfma; glhip_softmin_x32.h: fwd_tile), packed on the fly
    and from pre-packed records; fp32 and bf16 clouds, with and without column splits, against the C oracle on the same ranges."""
    rng = np.random.default_rng(41 + D)
    N, M = 2300, 2600
    x, y, h = _clouds(43, N, M, D)
    rg, tup, _, keep, ri = _random_ranges(rng, N, M, 80, 90, 0.3, cuda)
    rg.small_i = True          # the launch hint kernel_truncation sets (GLHIP_FLAG_SMALL_ROW_BLOCKS); blocks of up to ~150 rows are fine too
    xt, yt = _t(x, cuda), _t(y, cuda)
    if half:
        xt, yt = xt.bfloat16(), yt.bfloat16()
        x, y = xt.float().cpu().numpy(), yt.float().cpu().numpy()
    eps = 0.02
    ref = oracle_c.softmin(eps, x, y, h, 2, ranges=tup)
    live = np.isfinite(ref)
    assert (~live).sum() == ri[0, 1] - ri[0, 0]          # the row block with nothing to reduce over
    for flags in (0, hip.FLAG_NO_SPLIT, hip.FLAG_PREPACK, hip.FLAG_PREPACK | hip.FLAG_NO_SPLIT):
        out = hip.softmin(eps, xt, yt, _t(h, cuda), p=2, ranges=rg, flags=flags).cpu().numpy()
        assert np.isposinf(out[~live]).all(), flags
        assert np.abs(out[live] - ref[live]).max() < 1.2e-6 + 2e-6 * np.abs(ref[live]).max(), flags


def test_softmin_block_sparse_prepacked_path(cuda):
    """>= 5e8 nominal pairs: the block-sparse launch copies pre-packed column records (global centre, per-column layout,
    tiles starting at arbitrary columns) instead of packing per workgroup.  Against the oracle and the 16x16x32 kernel."""
    rng = np.random.default_rng(5)
    N, M, D = 23000, 24000, 3
    x, y, h = _clouds(29, N, M, D)
    x, y = np.sort(x, axis=0), np.sort(y, axis=0)           # loosely cluster-sorted along every axis
    rg, tup, _, keep, ri = _random_ranges(rng, N, M, 57, 61, 0.12, cuda)
    eps = 0.03**2
    ref = oracle_c.softmin(eps, x, y, h, 2, ranges=tup)
    live = np.ones(N, bool)
    live[ri[0, 0]:ri[0, 1]] = False
    live &= np.isfinite(ref)
    outs = {}
    for flags in (0, hip.FLAG_XDL16, hip.FLAG_NO_SPLIT):
        outs[flags] = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(h, cuda), p=2, ranges=rg, flags=flags).cpu().numpy()
        assert np.abs(outs[flags][live] - ref[live]).max() < 4e-7 * D + 2e-6 * np.abs(ref[live]).max(), flags
    assert np.isposinf(outs[0][~live]).all()


def test_block_sparse_leftover_row_tiles_are_carried(cuda):
    """f16 x 2 layout, 4-wavefront workgroups (round 6): a row block of nt > 4 row tiles of 32 is cut into nt / 4 chunks of exactly 4
    tiles and its nt mod 4 leftover tiles ride with the workgroups of the first chunks, each shared column-group-wise by the four
    wavefronts (glhip_softmin_x32.h, build_row_chunks_kernel share mode).  Row blocks of every nt from 1 to 23 with ragged last
    tiles — 129 rows (1 carried row), 160, 161 (a carried tile AND a trailing chunk of one row), 225 (5 - 7 tiles: W = 1 < rl),
    353 (11 tiles: W = 2 < rl = 3), 416, 455, 530, 737 — on dense-ish and very sparse column patterns, with / without column
    splits, pre-packed or not, a lazy-max stress vector, against the C oracle; and the 16x16x32 kernel on the same ranges."""
    rng = np.random.default_rng(61)
    sizes_i = np.array([129, 5, 160, 161, 33, 225, 128, 256, 257, 353, 300, 416, 455, 96, 530, 640, 737, 1, 200], dtype=np.int64)
    sizes_j = np.array([300, 40, 1, 700, 90, 513, 31, 260, 1200, 64], dtype=np.int64)
    N, M, D = int(sizes_i.sum()), int(sizes_j.sum()), 3
    x, y, h = _clouds(67, N, M, D)
    ends_i, ends_j = np.cumsum(sizes_i), np.cumsum(sizes_j)
    ri = np.stack([ends_i - sizes_i, ends_i], 1).astype(np.int32)
    rj = np.stack([ends_j - sizes_j, ends_j], 1).astype(np.int32)
    eps = 0.02
    h_spiky = (np.arange(M) // 64 * 48.0).astype(np.float32)      # every tile overflows its speculative pass and is redone exactly
    h_spiky[M // 2:] -= 3000.0
    h_spiky[-3] = 5000.0
    for density in (0.6, 0.12):
        keep = rng.random((len(sizes_i), len(sizes_j))) < density
        keep[:, 2] = True                                   # every row block reduces over something (a single column here)
        keep[3, :] = False
        keep[3, 6] = True                                   # 161 rows against 31 columns: fewer column groups than wavefronts
        rg = from_matrix(torch.from_numpy(ri).to(cuda), torch.from_numpy(rj).to(cuda), torch.from_numpy(keep).to(cuda))
        tup = tuple(t.cpu().numpy() for t in (rg.ranges_i, rg.slices_i, rg.redranges_j))
        for hv in (h, h_spiky):
            ref = oracle_c.softmin(eps, x, y, hv, 2, ranges=tup)
            for flags in (hip.FLAG_F16X2, hip.FLAG_F16X2 | hip.FLAG_NO_SPLIT, hip.FLAG_F16X2 | hip.FLAG_PREPACK, hip.FLAG_XDL16):
                out = hip.softmin(eps, _t(x, cuda), _t(y, cuda), _t(hv, cuda), p=2, ranges=rg, flags=flags).cpu().numpy()
                assert np.isfinite(out).all(), flags
                err = np.abs(out - ref)
                assert err.max() < 2e-6 + 3e-6 * np.abs(ref).max(), (flags, density, int(err.argmax()))


def test_fused_ends_of_a_small_loss(cuda):
    """``glhip_log_weights`` and ``glhip_sinkhorn_cost`` (round 6): the three multi-tensor launches of ``log_weights`` and the seven
    elementwise launches of the balanced loss formula (sinkhorn_divergence.py:61-65,171-199) as one kernel each.  Values against the
    torch expressions they replace; first and second derivatives of the cost through autograd (the formula is bilinear: the backward
    pass is differentiable torch code); shared and per-item weights, with and without debiasing, batched and not."""
    import geomloss_amd.sinkhorn_divergence as sd
    g = torch.Generator().manual_seed(3)
    ws = [torch.rand(n, generator=g).to(cuda) for n in (700, 1, 4097)] + [torch.rand(3, 50, generator=g).to(cuda)]
    ws[0][::7] = 0.0
    ws[0][3] = -1.0
    ws[2][5] = float("nan")
    got = hip.log_weights_raw(ws)
    for w, l in zip(ws, got):
        want = w.clamp_min(0).log().clamp_min(-100000.0)
        assert l.shape == w.shape and torch.equal(torch.isnan(l), torch.isnan(want))
        assert torch.equal(torch.nan_to_num(l, nan=0.0), torch.nan_to_num(want, nan=0.0)) or (torch.nan_to_num(l) - torch.nan_to_num(want)).abs().max() < 1e-6
    assert all(torch.equal(a_, b_) for a_, b_ in zip(sd.log_weights_many(ws[:2]), got[:2]))      # the hook takes the fused path

    for batch, debias, shared in ((True, True, False), (True, False, False), (False, True, False), (True, True, True)):
        B, N, M = (4, 300, 450) if batch else (1, 300, 450)
        shp = (lambda n: (B, n)) if batch else (lambda n: (n,))
        mk = lambda s_: (torch.randn(s_, generator=g).to(cuda) * 0.1).requires_grad_(True)  # noqa: E731
        f_ba, f_aa, g_ab, g_bb = mk(shp(N)), mk(shp(N)), mk(shp(M)), mk(shp(M))
        a = (torch.rand((N,) if shared else shp(N), generator=g).to(cuda) / N).requires_grad_(True)
        b = (torch.rand((M,) if shared else shp(M), generator=g).to(cuda) / M).requires_grad_(True)
        fused = hip.sinkhorn_cost_fused(a, f_ba, f_aa if debias else None, b, g_ab, g_bb if debias else None, batch)
        dd = lambda t: t.double()  # noqa: E731
        fa = dd(f_ba) - dd(f_aa) if debias else dd(f_ba)
        gb = dd(g_ab) - dd(g_bb) if debias else dd(g_ab)
        want = (dd(a) * fa).sum(-1) + (dd(b) * gb).sum(-1)
        assert fused.shape == want.shape and relerr(fused.detach().cpu().numpy(), want.detach().cpu().numpy()) < 2e-7
        ins = [a, f_ba, b, g_ab] + ([f_aa, g_bb] if debias else [])
        v = torch.randn(want.shape, generator=g).to(cuda)
        g1 = torch.autograd.grad((fused * v).sum(), ins, create_graph=True)
        g2 = torch.autograd.grad((want * v.double()).sum(), ins, create_graph=True)
        for u, w_ in zip(g1, g2):
            assert u.shape == w_.shape and relerr(u.detach().cpu().numpy(), w_.detach().cpu().numpy()) < 1e-6
        # second order: d/df_ba of <grad_a, r> — the mixed derivative a bilinear form has
        r = torch.randn(a.shape, generator=g).to(cuda)
        h1, = torch.autograd.grad((g1[0] * r).sum(), [f_ba])
        h2, = torch.autograd.grad((g2[0] * r.double()).sum(), [f_ba])
        assert relerr(h1.cpu().numpy(), h2.cpu().numpy()) < 1e-6
    # glhip_bounding_box: the reductions of max_diameter in one launch — exact, NaN coordinates propagate as in torch.aminmax
    import geomloss_amd.sinkhorn_divergence as sdv
    for D, nx, ny, dt in ((3, 2000, 1700, torch.float32), (1, 5, 7000, torch.float32), (16, 300, 1, torch.float32), (7, 999, 1001, torch.bfloat16)):
        xb, yb = (torch.randn(nx, D, generator=g).to(cuda).to(dt)), (torch.randn(ny, D, generator=g).to(cuda).to(dt) * 2 + 1)
        lo, hi = hip.bounding_box(xb, yb)
        z = torch.cat((xb, yb)).float()
        assert torch.equal(lo, z.min(0)[0]) and torch.equal(hi, z.max(0)[0])
        if dt == torch.float32:
            assert hip.bounding_box_applies(xb, yb)
            assert sdv.max_diameter(xb, yb) == (z.max(0)[0] - z.min(0)[0]).norm().item()      # the same float32 norm of the same extents
    xb[3, 0] = float("nan")
    lo, hi = hip.bounding_box(xb, yb)
    assert torch.isnan(lo[0]) and torch.isnan(hi[0]) and not torch.isnan(lo[1:]).any()
    # the hook of the loss formula
    f_ba, f_aa, g_ab, g_bb = (torch.randn(2, 64, generator=g).to(cuda) for _ in range(4))
    a = b = torch.full((2, 64), 1 / 64, device=cuda)
    out = sd.sinkhorn_cost(0.01, None, a, b, f_aa, g_bb, g_ab, f_ba, batch=True, debias=True)
    assert relerr(out.cpu().numpy(), ((a * (f_ba - f_aa)).sum(1) + (b * (g_ab - g_bb)).sum(1)).cpu().numpy()) < 1e-6


def test_block_sparse_very_uneven_row_blocks(cuda):
    """Row blocks of 1 ... 6000 rows (voxel clusters of a cloud sampled on a surface look like this): the launch cuts them
    into row chunks (build_row_chunks_kernel), one workgroup each.  Forward, gradient, gaussian product and gradient against
    the C oracle; with and without a workspace (no table: one workgroup per row block)."""
    rng = np.random.default_rng(31)
    sizes_i = np.array([1, 6000, 2, 1, 700, 3, 257, 256, 1, 1300, 5, 64], dtype=np.int64)
    sizes_j = np.array([900, 1, 1, 4000, 2, 300, 31, 1, 1200], dtype=np.int64)
    N, M, D = int(sizes_i.sum()), int(sizes_j.sum()), 3
    x, y, h = _clouds(37, N, M, D)
    ends_i, ends_j = np.cumsum(sizes_i), np.cumsum(sizes_j)
    ri = np.stack([ends_i - sizes_i, ends_i], 1).astype(np.int32)
    rj = np.stack([ends_j - sizes_j, ends_j], 1).astype(np.int32)
    keep = rng.random((len(sizes_i), len(sizes_j))) < 0.5
    keep[:, 3] = True                                     # every row block sees the big column block
    rg = from_matrix(torch.from_numpy(ri).to(cuda), torch.from_numpy(rj).to(cuda), torch.from_numpy(keep).to(cuda))
    tup = tuple(t.cpu().numpy() for t in (rg.ranges_i, rg.slices_i, rg.redranges_j))
    eps, blur = 0.02, 0.1
    g = rng.standard_normal(N).astype(np.float32)
    v = (rng.random(M) / M).astype(np.float32)
    ref = oracle_c.softmin(eps, x, y, h, 2, ranges=tup)
    refg = oracle_c.softmin_grad_x(eps, x, y, h, g, 2, ranges=tup)
    refk = oracle_c.kconv("gaussian", x, y, v, blur, ranges=tup)
    refkg = oracle_c.kconv_grad_x("gaussian", x, y, v, g, blur, ranges=tup)
    for flags in (0, hip.FLAG_NO_SPLIT, hip.FLAG_NO_MFMA, hip.FLAG_PREPACK, hip.FLAG_XDL16):
        xt = _t(x, cuda).requires_grad_(True)
        out = hip.softmin(eps, xt, _t(y, cuda), _t(h, cuda), p=2, ranges=rg, flags=flags)
        assert np.abs(out.detach().cpu().numpy() - ref).max() < 1.2e-6 + 2e-6 * np.abs(ref).max(), flags
        (gx,) = torch.autograd.grad(out, [xt], grad_outputs=_t(g, cuda))
        assert relerr(gx.cpu().numpy(), refg) < 1e-5, flags
        xt = _t(x, cuda).requires_grad_(True)
        k = hip.kernel_conv("gaussian", xt, _t(y, cuda), _t(v, cuda), blur, ranges=rg, flags=flags)
        assert relerr(k.detach().cpu().numpy(), refk) < 1e-4, flags
        (gk,) = torch.autograd.grad(k, [xt], grad_outputs=_t(g, cuda))
        assert relerr(gk.cpu().numpy(), refkg) < 1e-4, flags
    # p = 1 / laplacian: the VALU operators go through the same table
    out1 = hip.softmin(0.05, _t(x, cuda), _t(y, cuda), _t(h, cuda), p=1, ranges=rg).cpu().numpy()
    assert relerr(out1, oracle_c.softmin(0.05, x, y, h, 1, ranges=tup)) < 3e-6
    kl = hip.kernel_conv("laplacian", _t(x, cuda), _t(y, cuda), _t(v, cuda), blur, ranges=rg).cpu().numpy()
    assert relerr(kl, oracle_c.kconv("laplacian", x, y, v, blur, ranges=tup)) < 3e-6
    # no workspace at all (raw C-ABI call): the kernels fall back to one workgroup per row block
    lib = hip.load_library()
    xt, yt, ht = _t(x, cuda)[None].contiguous(), _t(y, cuda)[None].contiguous(), _t(h, cuda)[None].contiguous()
    o = torch.empty((1, N), device=cuda)
    rc = lib.glhip_softmin_fwd(xt.data_ptr(), yt.data_ptr(), ht.data_ptr(), o.data_ptr(), 1, N, M, D, eps, 2, hip.F32,
                               *rg.c_args(), None, 0, 0, None)
    torch.cuda.synchronize()
    assert rc == 0 and np.abs(o.cpu().numpy()[0] - ref).max() < 1.2e-6 + 2e-6 * np.abs(ref).max()


KINDS = ["gaussian", "laplacian", "energy"]


def test_gaussian_product_many_columns(cuda):
    """M >= 65536: XCD-aware grid with an adaptive number of column splits + pre-packed columns (32x32x16 kernel)."""
    N, M, D = 900, 70001, 3
    x, y, v = _clouds(77, N, M, D)
    v = np.abs(v) / M
    v[::7] *= -1.0                                          # signed weights are legal
    blur = 0.07
    ref = oracle_c.kconv("gaussian", x, y, v, blur)
    bound = oracle_c.kconv("gaussian", x, y, np.abs(v), blur)
    tol = 3e-6 * np.abs(ref).max() + 2.4e-7 * D / blur**2 * np.abs(bound).max()   # as in test_kernel_conv_vs_oracle
    for flags in (0, hip.FLAG_XDL16, hip.FLAG_NO_MFMA):
        out = hip.kernel_conv("gaussian", _t(x, cuda), _t(y, cuda), _t(v, cuda), blur, flags=flags).cpu().numpy()
        assert np.abs(out - ref).max() < tol, flags


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("N,M,D", [(300, 257, 3), (1030, 2100, 2), (200, 300, 1), (150, 170, 6)])
def test_kernel_conv_vs_oracle(cuda, kind, N, M, D):
    x, y, v = _clouds(31 + N, N, M, D)
    blur = 0.2
    ref = oracle_c.kconv(kind, x, y, v, blur)
    out = hip.kernel_conv(kind, _t(x, cuda), _t(y, cuda), _t(v, cuda), blur, flags=hip.FLAG_NO_MFMA).cpu().numpy()
    assert relerr(out, ref) < 3e-6
    # default path (gaussian, D <= 3: expanded exponent on the matrix cores): each kernel value carries a relative
    # error ~2^-22 * |x - c|^2 / blur^2 with random sign
    out = hip.kernel_conv(kind, _t(x, cuda), _t(y, cuda), _t(v, cuda), blur).cpu().numpy()
    bound = oracle_c.kconv(kind, x, y, np.abs(v), blur) if kind == "gaussian" else None
    tol = 3e-6 * np.abs(ref).max() + (2.4e-7 * D / blur**2 * np.abs(bound).max() if kind == "gaussian" else 0)
    assert np.abs(out - ref).max() < tol
    for flags in (hip.FLAG_NO_SPLIT, hip.FLAG_PREPACK, hip.FLAG_XDL16):
        out = hip.kernel_conv(kind, _t(x, cuda), _t(y, cuda), _t(v, cuda), blur, flags=flags).cpu().numpy()
        assert np.abs(out - ref).max() < tol, flags


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("D", [3, 2, 5])
def test_kernel_conv_grads_vs_oracle(cuda, kind, D):
    N, M = 310, 270
    x, y, v = _clouds(41 + D, N, M, D)
    y[:7] = x[:7]    # coincident points: |x-y| = 0 must give a zero direction, not NaN
    blur = 0.15
    g = np.random.default_rng(4).standard_normal(N).astype(np.float32)
    # default path: the gaussian kernel runs on the matrix cores up to D = 16 (expanded exponent: ~1e-5), the others on explicit differences
    for flags, tol in ((hip.FLAG_NO_MFMA, 5e-6), (0, 5e-6 if kind != "gaussian" else 1e-4)):
        xt, yt, vt = (_t(a, cuda).requires_grad_(True) for a in (x, y, v))
        out = hip.kernel_conv(kind, xt, yt, vt, blur, flags=flags)
        gx, gy, gv = torch.autograd.grad(out, [xt, yt, vt], grad_outputs=_t(g, cuda))
        assert relerr(gx.cpu().numpy(), oracle_c.kconv_grad_x(kind, x, y, v, g, blur)) < tol
        assert relerr(gy.cpu().numpy(), oracle_c.kconv_grad_x(kind, y, x, g, v, blur)) < tol
        assert relerr(gv.cpu().numpy(), oracle_c.kconv(kind, y, x, g, blur)) < tol


@pytest.mark.parametrize("D", [3, 2])
def test_gaussian_gradient_family_in_compact_row_order(cuda, monkeypatch, D):
    """Big dense gaussian launches of the kernels with one centre per workgroup (product + gradient, gradients, GRAD_FAMILY products)
    get their rows in compact order first (hip._gauss_compact_rows; threshold lowered here): same results in the caller's order,
    closer to the oracle — the exponent is assembled from terms of size |x - c| |y - c| / blur^2, c the workgroup's first row."""
    N, M = 5000, 4100
    x, y, v = _clouds(77 + D, N, M, D)
    v = np.abs(v) / M
    blur = 0.03                         # diam / blur ~ 50: the regime where the float32 exponent error shows
    g = np.random.default_rng(6).standard_normal(N).astype(np.float32)
    ref = dict(out=oracle_c.kconv("gaussian", x, y, v, blur), gx=oracle_c.kconv_grad_x("gaussian", x, y, v, g, blur),
               gy=oracle_c.kconv_grad_x("gaussian", y, x, g, v, blur), gv=oracle_c.kconv("gaussian", y, x, g, blur))
    sorts = []
    orig = hip.compact_order
    monkeypatch.setattr(hip, "compact_order", lambda *a, **k: (sorts.append(1), orig(*a, **k))[1])
    err = {}
    for tag, thr in (("scattered", 1e30), ("compact", 0.0)):
        monkeypatch.setattr(hip, "_GAUSS_SORT_MIN_PAIRS", thr)
        for fused in (True, False):
            hip.set_kernel_grad_fusion(fused)
            try:
                n0 = len(sorts)
                xt, yt, vt = (_t(a, cuda).requires_grad_(True) for a in (x, y, v))
                out = hip.kernel_conv("gaussian", xt, yt, vt, blur, flags=hip.FLAG_GRAD_FAMILY)
                gx, gy, gv = torch.autograd.grad(out, [xt, yt, vt], grad_outputs=_t(g, cuda))
                assert (len(sorts) - n0 > 0) == (tag == "compact")
            finally:
                hip.set_kernel_grad_fusion(True)
            got = dict(out=out.detach(), gx=gx, gy=gy, gv=gv)
            err[tag, fused] = {k: relerr(t.cpu().numpy(), ref[k]) for k, t in got.items()}
            print(tag, fused, {k: f"{e:.1e}" for k, e in err[tag, fused].items()})
            assert max(err[tag, fused].values()) < (8e-5 if tag == "compact" else 1e-3), (tag, fused, err[tag, fused])
    for fused in (True, False):
        for k in ("out", "gx", "gy", "gv"):         # 5000 points: a workgroup's 256 rows still span a third of the cloud; 1e6: 1 / 26
            assert err["compact", fused][k] < 0.5 * err["scattered", fused][k], (k, fused, err)
    monkeypatch.setattr(hip, "_GAUSS_SORT_MIN_PAIRS", 0.0)
    n0 = len(sorts)
    hip.kernel_conv("gaussian", _t(x, cuda), _t(y, cuda), _t(v, cuda), blur)                    # forward family: global centre, no sort
    xb = _t(np.stack([x, x]), cuda).requires_grad_(True)
    hip.kernel_conv("gaussian", xb, _t(np.stack([y, y]), cuda), _t(np.stack([v, v]), cuda), blur)     # batches: no sort
    assert len(sorts) == n0


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("N,M,D,B", [(310, 270, 3, None), (1030, 70_001, 3, None), (257, 300, 2, 3), (90, 80, 1, None)])
def test_kernel_product_and_gradient_in_one_pass(cuda, kind, N, M, D, B):
    """glhip_kernel_conv_fwd_grad (what the autograd forward runs when x requires gradients) against the two separate
    reductions and the oracle: same product, same gradient, for dense / many-column / batched / bf16 launches."""
    x, y, v = _clouds(53 + N, N, M, D, B=B)
    v = np.abs(v) / M
    if kind != "gaussian":
        y[..., :5, :] = x[..., :5, :]          # coincident points: the clamp of utils.py:61 and a zero direction
    blur = 0.12
    g = np.random.default_rng(5).standard_normal(x.shape[:-1]).astype(np.float32)
    res = {}
    for fused in (True, False):
        hip.set_kernel_grad_fusion(fused)
        try:
            xt = _t(x, cuda).requires_grad_(True)
            out = hip.kernel_conv(kind, xt, _t(y, cuda), _t(v, cuda), blur)
            (gx,) = torch.autograd.grad(out, [xt], grad_outputs=_t(g, cuda))
            res[fused] = (out.detach().cpu().numpy(), gx.cpu().numpy())
        finally:
            hip.set_kernel_grad_fusion(True)
    assert relerr(res[True][0], res[False][0]) < 2e-5 and relerr(res[True][1], res[False][1]) < 2e-5
    tol = 1e-4 if kind == "gaussian" else 5e-6
    if B is None:
        assert relerr(res[True][0], oracle_c.kconv(kind, x, y, v, blur)) < tol
        assert relerr(res[True][1], oracle_c.kconv_grad_x(kind, x, y, v, g, blur)) < tol
    # the explicit-difference kernels: the product of the fused mode IS the product mode, bit for bit (the three terms of a
    # kernel norm must share their rounding, kernel_samples._kernel_operators)
    x3, y3, v3 = (_t(a, cuda).reshape((1,) + a.shape if B is None else a.shape).contiguous() for a in (x, y, v))
    code = hip.KERNEL_KINDS[kind]
    fl = hip.FLAG_NO_MFMA | hip.FLAG_GRAD_FAMILY
    o_fused, _ = hip.kernel_conv_fwd_grad_raw(code, x3, y3, v3, blur, flags=fl)
    o_plain = hip.kernel_conv_fwd_raw(code, x3, y3, v3, blur, flags=fl)
    assert torch.equal(o_fused, o_plain)
    # bf16 clouds
    xb, yb = _t(x, cuda).bfloat16().requires_grad_(True), _t(y, cuda).bfloat16()
    out = hip.kernel_conv(kind, xb, yb, _t(v, cuda), blur)
    (gb,) = torch.autograd.grad(out, [xb], grad_outputs=_t(g, cuda))
    if B is None:
        ref = oracle_c.kconv_grad_x(kind, xb.detach().float().cpu().numpy(), yb.float().cpu().numpy(), v, g, blur)
        assert relerr(gb.float().cpu().numpy(), ref) < 2.0 ** -7
    with pytest.raises(NotImplementedError):      # one-pass kernels: D <= 3 (gaussian: D <= 16, tests/test_xd_kernels_gpu.py)
        hip.kernel_conv_fwd_grad_raw(code, _t(np.zeros((1, 8, 17), np.float32), cuda), _t(np.zeros((1, 8, 17), np.float32), cuda),
                                     _t(np.zeros((1, 8), np.float32), cuda), blur)


@pytest.mark.parametrize("kind", KINDS)
def test_kernel_conv_block_sparse(cuda, kind):
    rng = np.random.default_rng(19)
    N, M, D = 1900, 2100, 3
    x, y, v = _clouds(29, N, M, D)
    rg, tup, tup_t, keep, ri = _random_ranges(rng, N, M, 8, 7, 0.5, cuda)
    blur = 0.3
    g = rng.standard_normal(N).astype(np.float32)
    for flags, tol in ((hip.FLAG_NO_MFMA, 3e-6), (0, 3e-6 if kind != "gaussian" else 1e-4),
                       (hip.FLAG_PREPACK, 3e-6 if kind != "gaussian" else 1e-4)):
        out = hip.kernel_conv(kind, _t(x, cuda), _t(y, cuda), _t(v, cuda), blur, ranges=rg, flags=flags).cpu().numpy()
        assert relerr(out, oracle_c.kconv(kind, x, y, v, blur, ranges=tup)) < tol
        # transposed pattern (K^T @ g)
        out_t = hip.kernel_conv(kind, _t(y, cuda), _t(x, cuda), _t(g, cuda), blur, ranges=rg.t(), flags=flags).cpu().numpy()
        assert relerr(out_t, oracle_c.kconv(kind, y, x, g, blur, ranges=tup_t)) < tol


@pytest.mark.parametrize("D", [4, 7, 16, 33])
@pytest.mark.parametrize("p", [2, 1])
def test_generic_dimension_softmin(cuda, D, p):
    N, M = 270, 310
    x, y, h = _clouds(51 + D, N, M, D)
    eps = 0.3
    ref = oracle_c.softmin(eps, x, y, h, p)
    xt = _t(x, cuda).requires_grad_(True)
    out = hip.softmin(eps, xt, _t(y, cuda), _t(h, cuda), p=p)
    assert relerr(out.detach().cpu().numpy(), ref) < 3e-6
    g = np.random.default_rng(6).standard_normal(N).astype(np.float32)
    (gx,) = torch.autograd.grad(out, [xt], grad_outputs=_t(g, cuda))
    assert relerr(gx.cpu().numpy(), oracle_c.softmin_grad_x(eps, x, y, h, g, p)) < 5e-6


def test_dense_softmin_vs_oracle(cuda):
    rng = np.random.default_rng(2)
    for (B, N, M) in [(2, 130, 1024), (1, 77, 333), (3, 5, 7)]:
        C = rng.random((B, N, M)).astype(np.float32) * 3
        h = rng.standard_normal((B, M)).astype(np.float32)
        for eps in (1.0, 0.01):
            ref = oracle_np.softmin_dense(eps, C.astype(np.float64), h.astype(np.float64))
            out = hip.softmin_dense(eps, _t(C, cuda), _t(h, cuda)).cpu().numpy()
            assert relerr(out, ref) < 2e-6


def test_c_abi_error_codes(cuda):
    lib = hip.load_library()
    x = torch.rand(10, 3, device=cuda)
    with pytest.raises(NotImplementedError):
        hip.softmin(0.1, x, x, torch.zeros(10, device=cuda), p=3)
    with pytest.raises(ValueError):
        hip.softmin(-1.0, x, x, torch.zeros(10, device=cuda), p=2)
    assert lib.glhip_version() >= 100


@pytest.mark.parametrize("N,M,D,B", [(700, 900, 3, None), (300, 2100, 2, None), (257, 255, 3, 3)])
@pytest.mark.parametrize("p", [2, 1])
def test_fused_sinkhorn_step_equals_unfused_composition(cuda, N, M, D, B, p):
    """glhip_sinkhorn_step == (prev + damping * softmin(eps, C, logw + pot/eps)) / 2, and the oracle."""
    x, y, logw = _clouds(61 + N, N, M, D, B=B)
    rng = np.random.default_rng(8)
    pot = (rng.standard_normal(logw.shape) * 0.05).astype(np.float32)
    prev = rng.standard_normal(x.shape[:-1]).astype(np.float32)
    eps, damping = 0.01, 0.8
    xt, yt = _t(x, cuda), _t(y, cuda)
    unfused = 0.5 * (_t(prev, cuda) + damping * hip.softmin(eps, xt, yt, _t(logw + pot / np.float32(eps), cuda), p=p))
    fused = hip.sinkhorn_step(eps, xt, yt, _t(logw, cuda), _t(pot, cuda), _t(prev, cuda), damping, p=p)
    assert (fused - unfused).abs().max().item() < 2e-6
    first = hip.sinkhorn_step(eps, xt, yt, _t(logw, cuda), None, None, damping, p=p)
    ref = damping * (oracle_c.softmin(eps, x, y, logw, p) if B is None else
                     np.stack([oracle_c.softmin(eps, x[b], y[b], logw[b], p) for b in range(B)]))
    assert np.abs(first.cpu().numpy() - ref).max() < 4e-7 * D + 2e-6 * np.abs(ref).max()
    # p = 1 beyond D = 3: fused on dense launches since round 5 (glhip_dist_xd.h), composed from the soft-min kernel on block-sparse ones
    x5, y5, z5 = torch.rand(10, 5, device=cuda), torch.rand(12, 5, device=cuda), torch.zeros(12, device=cuda)
    assert hip.fused_step_applies(5, 1) and not hip.fused_step_applies(5, 1, 0, True) and not hip.fused_step_applies(5, 1, hip.FLAG_NO_MFMA)
    prev5 = torch.rand(10, device=cuda)
    assert torch.allclose(hip.sinkhorn_step(eps, x5, y5, z5, None, None, 0.5, p=1), 0.5 * hip.softmin(eps, x5, y5, z5, p=1), rtol=0, atol=1e-6)
    assert torch.allclose(hip.sinkhorn_step(eps, x5, y5, z5, 0.3 * z5 + 0.01, prev5, 0.5, p=1),
                          0.5 * (prev5 + 0.5 * hip.softmin(eps, x5, y5, z5 + (0.3 * z5 + 0.01) / eps, p=1)), rtol=0, atol=1e-6)
    assert torch.allclose(hip.sinkhorn_step(eps, x5, y5, z5, None, None, 0.5, p=1, flags=hip.FLAG_NO_MFMA), 0.5 * hip.softmin(eps, x5, y5, z5, p=1),
                          rtol=0, atol=1e-5)


@pytest.mark.parametrize("fl", [0, hip.FLAG_F16X2], ids=["bf16x3", "f16x2"])
@pytest.mark.parametrize("N,M,D,B", [(700, 900, 3, None), (130, 2100, 2, None), (257, 255, 1, 3), (3000, 40, 3, None),
                                     (600, 700, 5, None), (257, 255, 9, 3), (300, 1100, 16, None)])      # D > 3: the xd kernel's multi launch (round 5)
@pytest.mark.parametrize("debias", [True, False])
@pytest.mark.parametrize("p", [2, 1])
def test_iter4_equals_four_fused_steps(cuda, N, M, D, B, debias, fl, p):
    """glhip_sinkhorn_iter4 (one launch per Sinkhorn iteration) == the simultaneous glhip_sinkhorn_step calls it replaces,
    for the initialisation and for an averaged update, with and without the two debiasing reductions."""
    x, y, _ = _clouds(90 + N, N, M, D, B=B)
    rng = np.random.default_rng(4)
    sh = (lambda n: (n,)) if B is None else (lambda n: (B, n))
    a_log = np.log(rng.random(sh(N)) + 0.1).astype(np.float32)
    b_log = np.log(rng.random(sh(M)) + 0.1).astype(np.float32)
    if p == 1 and fl:
        pytest.skip("the f16 x 2 layout is a p = 2 layout")
    eps, damping = (0.02 if p == 2 else 0.1), 0.9
    tol = 2e-6 if p == 2 else 4e-6      # p = 1: the iteration runs the MFMA distance kernel, the half-steps of D <= 3 explicit differences
    xt, yt, al, bl = _t(x, cuda), _t(y, cuda), _t(a_log, cuda), _t(b_log, cuda)
    step = lambda rows, cols, lw, pot, prev: hip.sinkhorn_step(eps, rows, cols, lw, pot, prev, damping, p=p, flags=fl)  # noqa: E731

    init = hip.sinkhorn_iter4(eps, xt, yt, al, bl, None, damping, debias, flags=fl, p=p)
    want = [step(xt, yt, bl, None, None), step(yt, xt, al, None, None)]
    if debias:
        want += [step(xt, xt, al, None, None), step(yt, yt, bl, None, None)]
    assert len(init) == len(want)
    for got, ref in zip(init, want):
        assert got.shape == ref.shape and (got - ref).abs().max().item() < tol
    ref_np = oracle_c.softmin(eps, *(a if B is None else a[0] for a in (x, y, b_log)), p) * damping
    assert np.abs((init[0] if B is None else init[0][0]).cpu().numpy() - ref_np).max() < 4e-7 * D + 2e-6 * np.abs(ref_np).max()

    f_ba, g_ab = want[0] + 0.01, want[1] - 0.02          # any old potentials
    f_aa, g_bb = (want[2] * 0.5, want[3] * 0.7) if debias else (None, None)
    old = (f_ba, g_ab, f_aa, g_bb) if debias else (f_ba, g_ab)
    new = hip.sinkhorn_iter4(eps, xt, yt, al, bl, old, damping, debias, flags=fl, p=p)
    want2 = [step(xt, yt, bl, g_ab, f_ba), step(yt, xt, al, f_ba, g_ab)]
    if debias:
        want2 += [step(xt, xt, al, f_aa, f_aa), step(yt, yt, bl, g_bb, g_bb)]
    for got, ref in zip(new, want2):
        assert (got - ref).abs().max().item() < tol
    # simultaneous updates: outputs may not alias inputs (raw entry point)
    lib = hip.load_library()
    xb, yb = (xt, yt) if B is not None else (xt[None], yt[None])
    rc = lib.glhip_sinkhorn_iter4(xb.data_ptr(), yb.data_ptr(), al.data_ptr(), bl.data_ptr(), f_ba.data_ptr(), g_ab.data_ptr(),
                                  None, None, f_ba.data_ptr(), g_ab.data_ptr(), None, None, xb.shape[0], N, M, D, eps, damping,
                                  2, 0, 0, None, 0, 0, None)
    assert rc == -1 and b"alias" in lib.glhip_last_error()


@pytest.mark.parametrize("N,M,D,B,p", [(700, 900, 3, None, 2), (257, 255, 1, 3, 2), (600, 700, 5, None, 2), (500, 300, 3, None, 1), (300, 400, 9, 2, 1)])
@pytest.mark.parametrize("debias", [True, False])
def test_anneal_equals_one_iter4_call_per_temperature(cuda, N, M, D, B, p, debias):
    """glhip_sinkhorn_anneal == the initialisation and one glhip_sinkhorn_iter4 call per temperature (bit for bit: same launches),
    the f16 x 2 layout switched on from the temperature given; final potentials and the inputs of the last iteration; error paths."""
    x, y, _ = _clouds(31 + N, N, M, D, B=B)
    rng = np.random.default_rng(9)
    sh = (lambda n: (n,)) if B is None else (lambda n: (B, n))
    al = _t(np.log(rng.random(sh(N)) + 0.1).astype(np.float32), cuda)
    bl = _t(np.log(rng.random(sh(M)) + 0.1).astype(np.float32), cuda)
    xt, yt = _t(x, cuda), _t(y, cuda)
    eps_list = [2.0, 2.0, 0.8, 0.3, 0.1, 0.04, 0.02]
    dampings = [1.0 / (1.0 + e / 0.7) for e in eps_list]
    min_eps = 0.2 if p == 2 else float("inf")         # f16 x 2 for the first four temperatures only (p = 2)
    ref_plan = hip.Iter4Plan(xt, yt, al, bl, debias, p=p)
    flag = lambda e: hip.FLAG_F16X2 if e >= min_eps else 0      # noqa: E731
    ref_plan.extra_flags = flag(eps_list[0])
    pots = ref_plan.run(eps_list[0], dampings[0], None)
    before = None
    for e, d in zip(eps_list, dampings):
        before = tuple(t.clone() for t in pots)
        ref_plan.extra_flags = flag(e)
        pots = tuple(t.clone() for t in ref_plan.run(e, d, pots))
    plan = hip.Iter4Plan(xt, yt, al, bl, debias, p=p)
    new, old = plan.anneal(eps_list, dampings, min_eps)
    assert len(new) == (4 if debias else 2)
    for got, want in zip(new + old, pots + before):
        assert got.shape == want.shape and torch.equal(got, want)
    # the plan goes on where the library call stopped: one more iteration from Python reads `new` and leaves it intact
    more = plan.run(0.02, 1.0, new)
    ref_plan.extra_flags = 0
    for got, want in zip(more, ref_plan.run(0.02, 1.0, pots)):
        assert torch.equal(got, want)
    for got, want in zip(new, pots):
        assert torch.equal(got, want)
    # error paths: a buffer that appears twice, a non-positive temperature
    lib = hip.load_library()
    ptrs = ctypes.c_void_p * 4
    bufs = [t.data_ptr() for t in plan.sets[0]] + [None] * (4 - len(plan.sets[0]))
    farr = ctypes.c_float * 2
    call = lambda s0, s1, eps: lib.glhip_sinkhorn_anneal(*plan.fixed, ptrs(*s0), ptrs(*s1), *plan.dims, farr(*eps), farr(1.0, 1.0), 2, p,      # noqa: E731
                                                         plan.dtype, None, 0, 0, 0.0, None)
    assert call(bufs, bufs, (1.0, 0.5)) == -1 and b"twice" in lib.glhip_last_error()
    other = [t.data_ptr() for t in plan.sets[1]] + [None] * (4 - len(plan.sets[1]))
    assert call(bufs, other, (1.0, 0.0)) == -1 and b"eps[1]" in lib.glhip_last_error()


@pytest.mark.parametrize("fl", [0, hip.FLAG_F16X2], ids=["bf16x3", "f16x2"])
@pytest.mark.parametrize("N,M,Nc,Mc,D,B", [(3000, 2500, 130, 90, 3, None), (257, 700, 33, 64, 2, None), (640, 511, 17, 40, 1, 3),
                                           (900, 1000, 70, 65, 6, None), (300, 40, 300, 7, 3, None)])
@pytest.mark.parametrize("debias", [True, False])
@pytest.mark.parametrize("p", [2, 1])
def test_extrapolate4_equals_four_softmins(cuda, N, M, Nc, Mc, D, B, debias, fl, p):
    """glhip_sinkhorn_extrapolate4 (the coarse-to-fine jump as one launch) == the four `extrapolate_samples` soft-mins of the fine
    points against the coarse measures (sinkhorn_samples.py:533-544) it replaces, and the float64 oracle on the first of them."""
    if p == 1 and fl:
        pytest.skip("the f16 x 2 layout is a p = 2 layout")
    x, y, _ = _clouds(11 + N, N, M, D, B=B)
    xc, yc, _ = _clouds(12 + Nc, Nc, Mc, D, B=B)
    rng = np.random.default_rng(8)
    sh = (lambda n: (n,)) if B is None else (lambda n: (B, n))
    a_log = np.log(rng.random(sh(Nc)) + 0.1).astype(np.float32)
    b_log = np.log(rng.random(sh(Mc)) + 0.1).astype(np.float32)
    b_log[..., 0] = -100000.0         # a massless cluster (sinkhorn_divergence.log_weights)
    eps, damping = (0.03 if p == 2 else 0.1), 0.8
    pots_np = [0.1 * rng.standard_normal(sh(n)).astype(np.float32) for n in (Nc, Mc, Nc, Mc)]      # f_ba, g_ab, f_aa, g_bb on the coarse clouds
    xt, yt, xct, yct, al, bl = (_t(v, cuda) for v in (x, y, xc, yc, a_log, b_log))
    pots = [_t(v, cuda) for v in pots_np][: 4 if debias else 2]
    got = hip.sinkhorn_extrapolate4(eps, xt, yt, xct, yct, al, bl, pots, damping, flags=fl, p=p)
    one = lambda rows, cols, lw, pot: damping * hip.softmin(eps, rows, cols, lw + pot / eps, p=p, flags=fl)     # noqa: E731
    want = [one(xt, yct, bl, pots[1]), one(yt, xct, al, pots[0])]
    if debias:
        want += [one(xt, xct, al, pots[2]), one(yt, yct, bl, pots[3])]
    assert len(got) == len(want)
    tol = 2e-6 if p == 2 else 4e-6
    for g, w in zip(got, want):
        assert g.shape == w.shape and (g - w).abs().max().item() < tol
    first = lambda a: a if B is None else a[0]      # noqa: E731
    ref = damping * oracle_c.softmin(eps, first(x), first(yc), first(b_log) + first(pots_np[1]) / eps, p)
    assert np.abs(first(got[0]).cpu().numpy() - ref).max() < 4e-7 * D + 2e-6 * np.abs(ref).max() + (1e-5 if p == 1 else 0)
    # error paths of the raw entry point: an empty coarse measure, a missing debiasing output
    lib = hip.load_library()
    xb, yb, xcb, ycb = (t if B is not None else t[None] for t in (xt, yt, xct, yct))
    args = lambda nc, out3: (xb.data_ptr(), yb.data_ptr(), xcb.data_ptr(), ycb.data_ptr(), al.data_ptr(), bl.data_ptr(),      # noqa: E731
                             pots[0].data_ptr(), pots[1].data_ptr(), None, None, got[0].data_ptr(), got[1].data_ptr(), out3, None,
                             xb.shape[0], N, M, nc, Mc, D, eps, damping, p, 0, None, 0, 0, None)
    assert lib.glhip_sinkhorn_extrapolate4(*args(0, None)) == -1 and b"empty coarse" in lib.glhip_last_error()
    assert lib.glhip_sinkhorn_extrapolate4(*args(Nc, got[0].data_ptr())) == -1 and b"go together" in lib.glhip_last_error()


@pytest.mark.parametrize("N,M,D,B", [(300, 257, 3, None), (1030, 70_001, 2, None), (257, 300, 1, 3)])
@pytest.mark.parametrize("p", [2, 1])
def test_hard_c_transform_vs_numpy(cuda, N, M, D, B, p):
    """glhip_cmin_fwd: min_j [C(x_i,y_j) - g_j] — dense, many columns (column splits + min-merge), batched, block-sparse."""
    x, y, g = _clouds(61 + N, N, M, D, B=B)
    C = oracle_np.cost_matrix(x.astype(np.float64), y.astype(np.float64), p)
    ref = (C - g.astype(np.float64)[..., None, :]).min(-1)
    out = hip.cmin(_t(x, cuda), _t(y, cuda), _t(g, cuda), p=p).cpu().numpy()
    assert out.shape == ref.shape and np.abs(out - ref).max() < 3e-6 * max(1.0, np.abs(ref).max())
    if B is None and M < 5000:
        rng = np.random.default_rng(1)
        rg, tup, _, keep, ri = _random_ranges(rng, N, M, 5, 6, 0.5, cuda)
        outs = hip.cmin(_t(x, cuda), _t(y, cuda), _t(g, cuda), p=p, ranges=rg).cpu().numpy()
        mask = np.zeros((N, M), bool)
        sl, red = tup[1], tup[2]
        for k, (r0, r1) in enumerate(tup[0]):
            for q in range(sl[k - 1] if k else 0, sl[k]):
                mask[r0:r1, red[q][0]:red[q][1]] = True
        refs = np.where(mask, C - g[None, :], np.inf).min(-1)
        assert np.isposinf(outs[ri[0, 0]:ri[0, 1]]).all()
        live = np.isfinite(refs)
        assert np.abs(outs[live] - refs[live]).max() < 3e-6 * max(1.0, np.abs(refs[live]).max())


def _clustered(seed, N, M, D, dev, voxel):
    """Two clouds sorted by voxel cluster + the block-sparse pattern of a keep-mask on their centroids."""
    from geomloss_amd import cluster
    g = torch.Generator().manual_seed(seed)
    x, y = torch.rand(N, D, generator=g).to(dev), torch.rand(M, D, generator=g).to(dev)
    _, _, xc, xs, rx, _ = cluster.clusterize_device(None, x, voxel)
    _, _, yc, ys, ry, _ = cluster.clusterize_device(None, y, voxel)
    rg = cluster.block_ranges_device("within", xc, yc, None, None, rx, ry, (3 * voxel) ** 2)
    tup = tuple(t.cpu().numpy() for t in (rg.ranges_i, rg.slices_i, rg.redranges_j))
    return xs, ys, rg, tup


@pytest.mark.parametrize("N,M,D", [(6000, 7000, 3), (3000, 2500, 2), (900, 1100, 1)])
def test_distance_reductions_on_the_matrix_cores_block_sparse(cuda, N, M, D):
    """GLHIP_FLAG_MFMA_DIST (glhip_dist_x32.h) on cluster-sorted clouds — what the multiscale backends launch for p = 1 /
    laplacian: soft-min (plain and fused half-step), laplacian and energy products against the C oracle with the same ranges
    and against the direct-difference VALU operators."""
    voxel = 0.12 if D == 3 else 0.05
    xs, ys, rg, tup = _clustered(7 + D, N, M, D, cuda, voxel)
    x, y = xs.cpu().numpy(), ys.cpu().numpy()
    rng = np.random.default_rng(3)
    h, v = rng.standard_normal(M).astype(np.float32), (rng.random(M) / M).astype(np.float32)
    v[::5] *= -1
    for eps in (0.05, 0.01):
        ref = oracle_c.softmin(eps, x, y, h, 1, ranges=tup)
        live = np.isfinite(ref)
        got = hip.softmin_fwd_raw(xs[None], ys[None], _t(h, cuda)[None], eps, 1, rg, hip.FLAG_MFMA_DIST)[0].cpu().numpy()
        valu = hip.softmin_fwd_raw(xs[None], ys[None], _t(h, cuda)[None], eps, 1, rg, 0)[0].cpu().numpy()
        assert np.abs(got[live] - ref[live]).max() < 3e-6 * max(1.0, np.abs(ref[live]).max()), eps
        assert np.abs(got[live] - valu[live]).max() < 3e-6 * max(1.0, np.abs(ref[live]).max())
        assert np.isposinf(got[~live]).all()
        # fused half-step: (prev + damping * softmin(eps, C, logw + pot/eps)) / 2
        pot, prev = rng.standard_normal(M).astype(np.float32) * 0.1, rng.standard_normal(N).astype(np.float32)
        step = hip.sinkhorn_step_raw(xs[None], ys[None], _t(h, cuda)[None], _t(pot, cuda)[None], _t(prev, cuda)[None], eps, 0.9, 1, rg,
                                     hip.FLAG_MFMA_DIST)[0].cpu().numpy()
        ref_s = 0.5 * (prev + 0.9 * oracle_c.softmin(eps, x, y, h + pot / eps, 1, ranges=tup))
        assert np.abs(step[live] - ref_s[live]).max() < 3e-6 * max(1.0, np.abs(ref_s[live]).max())
    for kind, code, blur in (("laplacian", hip.LAPLACIAN, 0.07), ("energy", hip.ENERGY, 1.0)):
        ref = oracle_c.kconv(kind, x, y, v, blur, ranges=tup)
        bound = oracle_c.kconv(kind, x, y, np.abs(v), blur, ranges=tup)
        got = hip.kernel_conv_fwd_raw(code, xs[None], ys[None], _t(v, cuda)[None], blur, rg, hip.FLAG_MFMA_DIST)[0].cpu().numpy()
        assert np.abs(got - ref).max() < 5e-6 * np.abs(bound).max(), kind
    # bf16 clouds go through the same kernel
    gb = hip.softmin_fwd_raw(xs[None].bfloat16(), ys[None].bfloat16(), _t(h, cuda)[None], 0.05, 1, rg, hip.FLAG_MFMA_DIST)[0]
    vb = hip.softmin_fwd_raw(xs[None].bfloat16(), ys[None].bfloat16(), _t(h, cuda)[None], 0.05, 1, rg, 0)[0]
    fin = torch.isfinite(vb)
    assert (gb[fin] - vb[fin]).abs().max().item() < 3e-6 * max(1.0, vb[fin].abs().max().item())


def test_distance_reductions_dense_large_launches_sort_their_rows(cuda):
    """Dense p = 1 soft-min / laplacian / energy launches of >= 5e8 pairs and >= 65536 rows: the Python layer voxel-sorts the rows,
    runs the matrix-core kernel on a "every row block x all columns" pattern and un-sorts the result.  Against the float64 oracle."""
    from oracle import oracle_torch64 as o64
    N, M = 70_000, 8_000
    g = torch.Generator().manual_seed(2)
    x, y = torch.rand(N, 3, generator=g).to(cuda), torch.rand(M, 3, generator=g).to(cuda)
    h = (torch.randn(M, generator=g) * 2).to(cuda)
    v = (torch.rand(M, generator=g) / M).to(cuda)
    for eps in (0.05, 0.005):
        ref = o64.softmin(eps, x, y, h, p=1, device=cuda)
        out = hip.softmin(eps, x, y, h, p=1)
        hip.set_distance_on_mfma(False)
        try:
            plain = hip.softmin(eps, x, y, h, p=1)
        finally:
            hip.set_distance_on_mfma(True)
        e_new, e_old = np.abs(out.cpu().numpy() - ref).max(), np.abs(plain.cpu().numpy() - ref).max()
        print(f"p=1 soft-min eps={eps}: max abs error matrix cores {e_new:.2e}, direct differences {e_old:.2e} (|f| <= {np.abs(ref).max():.2f})")
        # worst row = the one with the closest neighbour: ~2^-24 (rho + d)^2 / d with rho the voxel diagonal (glhip_dist_x32.h)
        assert e_new < 2e-6 * max(1.0, np.abs(ref).max())
        # one fused half-step of the loop takes the same route (plan cached on the row tensor)
        pot, prev = torch.randn(M, generator=g).to(cuda) * 0.1, torch.randn(N, generator=g).to(cuda)
        st = hip.sinkhorn_step(eps, x, y, h, pot, prev, 0.8, p=1)
        ref_s = 0.5 * (prev.double().cpu().numpy() + 0.8 * o64.softmin(eps, x, y, h + pot / eps, p=1, device=cuda))
        assert np.abs(st.cpu().numpy() - ref_s).max() < 3e-6 * max(1.0, np.abs(ref_s).max())
    for kind, blur in (("laplacian", 0.05), ("energy", 1.0)):
        ref = o64.kconv(kind, x, y, v, blur, device=cuda)
        out = hip.kernel_conv(kind, x, y, v, blur).cpu().numpy()
        print(f"{kind}: rel error {np.abs(out - ref).max() / np.abs(ref).max():.2e}")
        assert np.abs(out - ref).max() < 5e-6 * np.abs(ref).max()
    # degenerate bounding box (a planar cloud in 3-D) and a far-away cloud: the voxel size adapts, the result does not move
    for xx, yy in ((x * torch.tensor([1.0, 1.0, 0.0], device=cuda), y * torch.tensor([1.0, 1.0, 0.0], device=cuda)), (x + 500.0, y + 500.0)):
        got = hip.softmin(0.05, xx, yy, h, p=1)
        hip.set_distance_on_mfma(False)
        try:
            want = hip.softmin(0.05, xx, yy, h, p=1)
        finally:
            hip.set_distance_on_mfma(True)
        assert (got - want).abs().max().item() < 2e-6 * max(1.0, want.abs().max().item())
    # gradients still flow (VALU gradient kernels on the un-sorted cloud)
    xg = x.clone().requires_grad_(True)
    (gx,) = torch.autograd.grad(hip.softmin(0.05, xg, y, h, p=1).sum(), [xg])
    ref_g = o64.softmin_grad_x(0.05, x, y, h, np.ones(N), p=1, device=cuda)
    assert relerr(gx.cpu().numpy(), ref_g) < 2e-5


@pytest.mark.parametrize("N,M,D,B", [(700, 900, 3, None), (1030, 70_001, 3, None), (257, 300, 2, 3), (300, 200, 1, None)])
def test_softmin_value_and_gradient_in_one_pass(cuda, N, M, D, B):
    """glhip_softmin_fwd_grad: the exact soft-min and its row gradient from a GUESS of the soft-min and a margin — whatever the guess,
    as long as the truth lies within the margin.  Dense, many columns (splits + merge), batched, block-sparse, bf16."""
    x, y, h = _clouds(71 + N, N, M, D, B=B)
    eps = 0.02
    rng = np.random.default_rng(9)
    ref = oracle_np.softmin_points(eps, x.astype(np.float64), y.astype(np.float64), h.astype(np.float64), 2)
    refg = oracle_np.softmin_points_grad_x(eps, x.astype(np.float64), y.astype(np.float64), h.astype(np.float64), np.ones(x.shape[:-1]), 2)
    xt, yt, ht = (_t(a, cuda) if B is not None else _t(a, cuda)[None] for a in (x, y, h))
    for margin in (1e-4, 0.2, 0.45):             # up to 22 eps off
        guess = ref + rng.uniform(-margin, margin, ref.shape)
        gt = _t(guess.astype(np.float32), cuda) if B is not None else _t(guess.astype(np.float32), cuda)[None]
        out, unit = hip.softmin_fwd_grad_raw(xt.contiguous(), yt.contiguous(), ht.contiguous(), gt.contiguous(), margin * 1.01, eps)
        out, unit = out.cpu().numpy().reshape(ref.shape), unit.cpu().numpy().reshape(refg.shape)
        assert np.abs(out - ref).max() < 4e-7 * D + 3e-6 * np.abs(ref).max(), margin
        assert relerr(unit, refg) < 3e-5, margin
    if B is None and M < 5000:
        rg, tup, _, keep, ri = _random_ranges(rng, N, M, 5, 6, 0.5, cuda)
        refs = oracle_c.softmin(eps, x, y, h, 2, ranges=tup)
        live = np.isfinite(refs)
        guess = np.where(live, refs, 0.0) + rng.uniform(-0.1, 0.1, N)
        out, unit = hip.softmin_fwd_grad_raw(xt, yt, ht, _t(guess.astype(np.float32), cuda)[None], 0.11, eps, rg)
        out = out[0].cpu().numpy()
        assert np.isposinf(out[~live]).all() and np.abs(out[live] - refs[live]).max() < 1.2e-6 + 3e-6 * np.abs(refs[live]).max()
        refgs = oracle_c.softmin_grad_x(eps, x, y, h, np.ones(N, np.float32), 2, ranges=tup)
        assert relerr(unit[0].cpu().numpy()[live], refgs[live]) < 3e-5
        with pytest.raises(NotImplementedError):
            hip.softmin_fwd_grad_raw(xt, yt, ht, ht[:, :N] if M >= N else ht, 0.1, eps, flags=hip.FLAG_NO_MFMA)


def test_empty_clouds(cuda):
    """N = 0 returns an empty result; M = 0 is the reduction over the empty set (+inf potential, zero kernel sum)."""
    x, y0 = torch.rand(5, 3, device=cuda), torch.rand(0, 3, device=cuda)
    h0 = torch.zeros(0, device=cuda)
    assert hip.softmin(0.1, y0, x, torch.zeros(5, device=cuda)).shape == (0,)
    assert torch.isposinf(hip.softmin(0.1, x, y0, h0)).all()
    assert torch.isposinf(hip.softmin(0.1, x, y0, h0, flags=hip.FLAG_NO_MFMA)).all()
    assert (hip.kernel_conv("gaussian", x, y0, h0, 0.3) == 0).all()
    assert (hip.kernel_conv("energy", x, y0, h0, 0.3) == 0).all()


@pytest.mark.parametrize("N,M,D", [(6000, 7000, 3), (3000, 2500, 2), (900, 1100, 1)])
def test_distance_kernel_gradients_on_the_matrix_cores_block_sparse(cuda, N, M, D):
    """glhip_dist_grad_x32.h on cluster-sorted clouds (GLHIP_FLAG_MFMA_DIST): laplacian / energy product + unit gradient in one
    pass, gradient alone, and the FAMILY product (|.| = m rsq(m)) that accompanies them, against the C oracle with the same
    ranges.  A few coincident points: inside the clamp the product sees the floor, the gradient nothing."""
    voxel = 0.12 if D == 3 else 0.05
    xs, ys, rg, tup = _clustered(17 + D, N, M, D, cuda, voxel)
    ys[:5] = xs[:5]
    x, y = xs.cpu().numpy(), ys.cpu().numpy()
    rng = np.random.default_rng(5)
    v = (rng.random(M) / M).astype(np.float32)
    v[::5] *= -1
    g = rng.standard_normal(N).astype(np.float32)
    fl = hip.FLAG_MFMA_DIST
    for kind, code, blur in (("laplacian", hip.LAPLACIAN, 0.07), ("energy", hip.ENERGY, 1.0)):
        ref = oracle_c.kconv(kind, x, y, v, blur, ranges=tup)
        bound = oracle_c.kconv(kind, x, y, np.abs(v), blur, ranges=tup)
        refg = oracle_c.kconv_grad_x(kind, x, y, v, g, blur, ranges=tup)
        refu = oracle_c.kconv_grad_x(kind, x, y, v, np.ones(N), blur, ranges=tup)
        out, unit = hip.kernel_conv_fwd_grad_raw(code, xs[None], ys[None], _t(v, cuda)[None], blur, rg, fl)
        assert np.abs(out[0].cpu().numpy() - ref).max() < 5e-6 * np.abs(bound).max(), kind
        tolg = 2e-5 if D > 1 else 5e-5        # on a line, x_i S0 - S1 is the difference of two one-sided sums
        assert relerr(unit[0].cpu().numpy(), refu) < tolg, kind
        gx = hip.kernel_conv_bwd_x_raw(code, xs[None], ys[None], _t(v, cuda)[None], _t(g, cuda)[None], blur, rg, fl)
        assert relerr(gx[0].cpu().numpy(), refg) < tolg, kind
        fam = hip.kernel_conv_fwd_raw(code, xs[None], ys[None], _t(v, cuda)[None], blur, rg, fl | hip.FLAG_GRAD_FAMILY)[0]
        assert np.abs(fam.cpu().numpy() - ref).max() < 5e-6 * np.abs(bound).max(), kind
        # the product of the fused kernel and the FAMILY product share their arithmetic (same distances, |.| = m rsq(m)); only the
        # order of the sums differs
        assert (fam - out[0]).abs().max().item() < 2e-6 * np.abs(bound).max(), kind
        # without a workspace (no column splits, one workgroup per row block): raw C-ABI call
        lib = hip.load_library()
        o2, u2 = torch.empty((1, N), device=cuda), torch.empty((1, N, D), device=cuda)
        vt = _t(v, cuda)[None].contiguous()
        rc = lib.glhip_kernel_conv_fwd_grad(code, xs.data_ptr(), ys.data_ptr(), vt.data_ptr(), o2.data_ptr(), u2.data_ptr(), 1, N, M, D,
                                            blur, hip.F32, *rg.c_args(), None, 0, fl, None)
        torch.cuda.synchronize()
        assert rc == 0 and np.abs(o2[0].cpu().numpy() - ref).max() < 5e-6 * np.abs(bound).max()
        assert relerr(u2[0].cpu().numpy(), refu) < tolg


def test_distance_kernel_losses_dense_large_launches(cuda):
    """SamplesLoss("energy" | "laplacian", backend="online") with gradients at a size where the three products of the norm take the
    sorted matrix-core route (product + gradient for the terms in x, FAMILY product for the y-y term): loss, dL/dx, dL/da against
    the float64 oracle.  ``shift``: a loss of the size of its terms, everything at 1e-4.  Same law: the loss is 1e-5 (energy) to
    1e-3 (laplacian) of its three terms, each an fp32 sum — it is judged on the scale of those terms (1e-7: their rounding must
    be COMMON to cancel; the first version of the fused kernel lost 1.7e-5 of a product to a single running accumulator)."""
    from geomloss_amd import SamplesLoss
    from oracle import oracle_torch64 as o64
    N = M = 70_000
    g = torch.Generator().manual_seed(12)
    x, y0 = torch.rand(N, 3, generator=g).to(cuda), torch.rand(M, 3, generator=g).to(cuda)
    for name, blur, shift in (("energy", None, False), ("laplacian", 0.05, False), ("energy", None, True), ("laplacian", 0.05, True)):
        y = y0 * 0.6 + 0.3 if shift else y0
        ref, rgx, rga = o64.kernel_loss(name, x, y, blur=0.05 if blur is None else blur, grad=True, device=cuda)
        xg = x.clone().requires_grad_(True)
        a = torch.full((N,), 1.0 / N, device=cuda, requires_grad=True)
        b = torch.full((M,), 1.0 / M, device=cuda)
        kw = {} if blur is None else dict(blur=blur)
        L = SamplesLoss(name, backend="online", **kw)(a, xg, b, y)
        gx, ga = torch.autograd.grad(L, [xg, a])
        e = (abs(L.item() - ref) / abs(ref), relerr(gx.cpu().numpy(), rgx), relerr(ga.cpu().numpy(), rga))
        term = abs(float(o64.kconv(name, x, x, np.full(N, 1.0 / N), 0.05 if blur is None else blur, device=cuda).mean()))   # <a, K_xx a>
        print(f"{name} 7e4 shift={shift}: loss {L.item():.6e} oracle {ref:.6e} rel {e[0]:.2e} (terms ~{term:.1e}); dL/dx rel {e[1]:.2e}; "
              f"dL/da rel {e[2]:.2e}")
        assert e[1] < 1e-4
        if shift:
            assert e[0] < 1e-4 and e[2] < 1e-4
        else:
            assert abs(L.item() - ref) < 2e-7 * max(term, abs(ref))
