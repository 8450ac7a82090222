"""Pins the CPU oracle to the reference: every golden vector in tests/golden/ was produced by running
jeanfeydy/geomloss 0.3.1 (tensorized backend) — see tests/golden/make_golden.py.  CPU only."""

import re

import numpy as np
import pytest

from conftest import golden_cases, load_golden, relerr
import torch

from oracle import oracle_c, oracle_np, oracle_torch64

CPU = torch.device("cpu")


def _oracle_loss(rec, **extra):
    kw = dict(rec["kwargs"])
    name = kw.pop("loss")
    a, x, b, y = rec["a"], rec["x"], rec["b"], rec["y"]
    if name == "sinkhorn":
        return oracle_np.sinkhorn_loss(x, y, a, b, **kw, **extra)
    return oracle_np.kernel_loss(name, x, y, a, b, blur=kw.get("blur", 0.05), **extra)


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_loss_matches_reference_f64(name):
    rec = load_golden(name)
    assert relerr(_oracle_loss(rec), rec["loss_f64"]) < 1e-8


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_potentials_match_reference_f64(name):
    rec = load_golden(name)
    F, G = _oracle_loss(rec, potentials=True)
    assert relerr(F, rec["F_f64"]) < 1e-7 and relerr(G, rec["G_f64"]) < 1e-7


@pytest.mark.parametrize("name", [n for n in golden_cases() if "batch" not in n and "reach" not in n])
def test_oracle_closed_form_gradient_matches_reference_autograd(name):
    """The closed form the HIP backward kernels implement == autograd through the reference's dense code."""
    rec = load_golden(name)
    kw = dict(rec["kwargs"])
    loss = kw.pop("loss")
    a, x, b, y = rec["a"], rec["x"], rec["b"], rec["y"]
    if loss == "sinkhorn":
        _, gx, ga = oracle_np.sinkhorn_loss_and_grad(x, y, a, b, **kw)
        assert relerr(ga, rec["ga_f64"]) < 1e-7
    else:
        gx = oracle_np.kernel_loss_grad_x(loss, x, y, a, b, blur=kw.get("blur", 0.05))
    assert relerr(gx, rec["gx_f64"]) < 1e-7


def test_oracle_softmin_matches_reference_softmin_tensorized():
    rec = load_golden("softmin_tensorized")
    x, y, h = rec["x"], rec["y"], rec["h"]
    for p in (1, 2):
        for eps in (1.0, 0.05**p):
            ref = rec[f"softmin_p{p}_eps{eps:g}"]
            assert relerr(oracle_np.softmin_points(eps, x, y, h, p), ref) < 1e-10
            for b in range(x.shape[0]):
                assert relerr(oracle_c.softmin(eps, x[b], y[b], h[b], p), ref[b]) < 1e-10


def test_oracle_cfg1_baseline_config():
    """BASELINE.json configs[0]: SamplesLoss('sinkhorn', p=2, blur=.05), N=M=2000, 2D."""
    rec = load_golden("cfg1_n2000_d2")
    L = oracle_np.sinkhorn_loss(rec["x"], rec["y"], p=2, blur=0.05)
    assert abs(L - float(rec["loss_f64"])) / float(rec["loss_f64"]) < 1e-7
    # the reference's own fp32 result is 2e-6 away from its fp64 result on this input
    assert abs(float(rec["loss_f32"]) - float(rec["loss_f64"])) / float(rec["loss_f64"]) < 1e-5


def test_c_oracle_block_sparse_equals_masked_dense():
    rng = np.random.default_rng(0)
    N, M, D = 120, 150, 3
    x, y, h = rng.random((N, D)), rng.random((M, D)), rng.standard_normal(M)
    ri = np.array([[0, 50], [50, 120]], np.int32)
    red = np.array([[0, 30], [100, 150], [20, 60]], np.int32)
    sl = np.array([2, 3], np.int32)
    out = oracle_c.softmin(0.05, x, y, h, 2, ranges=(ri, sl, red))
    C = oracle_np.cost_matrix(x, y, 2)
    mask = np.zeros((N, M), bool)
    mask[0:50, 0:30] = mask[0:50, 100:150] = mask[50:120, 20:60] = True
    ref = oracle_np.softmin_dense(0.05, np.where(mask, C, np.inf), h)
    assert relerr(out, ref) < 1e-12


def test_multiscale_oracle_reduces_to_tensorized_without_truncation_effect():
    """With a jump after the last iteration the two-scale loop only extrapolates; with a huge `truncate`
    the fine level is dense.  Either way the oracle must stay close to the single-scale answer, and the
    kept fraction it reports must be in (0, 1]."""
    rng = np.random.default_rng(5)
    x, y = rng.random((700, 3)), rng.random((800, 3)) * 0.7 + 0.2
    a, b = np.full(700, 1 / 700), np.full(800, 1 / 800)
    single = oracle_np.sinkhorn_loss(x, y, p=2, blur=0.05, scaling=0.8)
    multi, info = oracle_np.sinkhorn_multiscale(a, x, b, y, p=2, blur=0.05, scaling=0.8, truncate=5, return_info=True)
    assert info["jumps"][0] < len(info["eps_list"]) - 1
    assert all(0 < k <= 1 for k in info["kept_fraction"]) and len(info["kept_fraction"]) == 3
    assert abs(multi - single) / abs(single) < 5e-3
    for tr in (3, 10):
        other = oracle_np.sinkhorn_multiscale(a, x, b, y, p=2, blur=0.05, scaling=0.8, truncate=tr)
        assert abs(other - multi) / abs(multi) < 1e-4


# ---- the chunked float64 torch oracle (full-size checks on the GPU box) is pinned here, on the CPU -------------------

def _torch64_loss(rec, **extra):
    kw = dict(rec["kwargs"])
    name = kw.pop("loss")
    a, x, b, y = rec["a"], rec["x"], rec["b"], rec["y"]
    if name == "sinkhorn":
        return oracle_torch64.sinkhorn_loss(x, y, a, b, device=CPU, **kw, **extra)
    return oracle_torch64.kernel_loss(name, x, y, a, b, blur=kw.get("blur", 0.05), device=CPU, **extra)


def _dimension(name):
    m = re.search(r"_d(\d+)", name)
    return int(m.group(1)) if m else 3


@pytest.mark.parametrize("name", [n for n in golden_cases() if "batch" not in n and _dimension(n) <= 3] + golden_cases(mid=True))
def test_torch64_losses_potentials_gradients_match_reference_f64(name):
    """Loss, potentials and closed-form gradients of the chunked oracle vs the reference's float64 outputs — including the
    mid-size cases (N = 8000) that the NumPy oracle is too slow for."""
    rec = load_golden(name)
    assert relerr(_torch64_loss(rec), rec["loss_f64"]) < 1e-8
    F, G = _torch64_loss(rec, potentials=True)
    assert relerr(F, rec["F_f64"]) < 1e-7 and relerr(G, rec["G_f64"]) < 1e-7
    if "reach" not in name:
        _, gx, ga = _torch64_loss(rec, grad=True)
        assert relerr(gx, rec["gx_f64"]) < 1e-7 and relerr(ga, rec["ga_f64"]) < 1e-7


@pytest.mark.parametrize("p", [2, 1])
def test_torch64_reductions_match_numpy_oracle(p):
    rng = np.random.default_rng(3)
    N, M, D = 900, 1100, 3
    x, y = rng.random((N, D)) + 5.0, rng.random((M, D)) * 0.8 + 5.1      # off-centre clouds: the expansion is centred
    h, g, v = rng.standard_normal(M), rng.standard_normal(N), rng.random(M)
    x[:7] = y[:7]                                                         # coincident points (clamp of utils.py:61)
    for eps in (0.7, 0.05**p):
        ref = oracle_np.softmin_points(eps, x, y, h, p)
        for kw in (dict(), dict(exact=True), dict(budget=50_000)):        # expanded / explicit differences / many chunks
            assert np.abs(oracle_torch64.softmin(eps, x, y, h, p, device=CPU, **kw) - ref).max() < 1e-11
        rows = np.array([5, 0, 899, 17])
        assert np.abs(oracle_torch64.softmin(eps, x, y, h, p, device=CPU, rows=rows) - ref[rows]).max() < 1e-11
        gref = oracle_np.softmin_points_grad_x(eps, x, y, h, g, p)
        assert np.abs(oracle_torch64.softmin_grad_x(eps, x, y, h, g, p, device=CPU, budget=70_000) - gref).max() < 1e-10
        assert np.abs(oracle_torch64.softmin_grad_x(eps, x, y, h, g, p, device=CPU, rows=rows) - gref[rows]).max() < 1e-10
    if p == 2:
        for kind, blur in (("gaussian", 0.1), ("laplacian", 0.2), ("energy", 0.05)):
            ref = oracle_c.kconv(kind, x, y, v, blur)
            assert relerr(oracle_torch64.kconv(kind, x, y, v, blur, device=CPU, budget=60_000), ref) < 1e-12
            gref = oracle_c.kconv_grad_x(kind, x, y, v, g, blur)
            assert relerr(oracle_torch64.kconv_grad_x(kind, x, y, v, g, blur, device=CPU, budget=60_000), gref) < 1e-11


def test_torch_port_matches_reference():
    """oracle/tensorized_torch.py — the PyTorch-CPU port timed as ``cpu_baseline`` by bench.py — against the reference's own
    fp32 tensorized outputs: BASELINE configs[0] exactly, and two 3-D golden cases (uniform weights)."""
    from oracle.tensorized_torch import sinkhorn_tensorized_cpu

    rec = load_golden("cfg1_n2000_d2")
    cnt = {}
    L = sinkhorn_tensorized_cpu(torch.from_numpy(rec["x"])[None], torch.from_numpy(rec["y"])[None], count=cnt).item()
    assert abs(L - float(rec["loss_f32"])) <= 2e-5 * float(rec["loss_f32"])     # fp32 vs fp32: summation order only
    assert abs(L - float(rec["loss_f64"])) <= 1e-4 * float(rec["loss_f64"])
    assert cnt["softmin_calls"] == 36                                            # 4 + 4 * 7 + 4 (SURVEY App. A, D=2 unit square)
    for name in ("sinkhorn_p2_d2", "sinkhorn_p2_scaling9"):                       # the uniform-weight golden cases
        rec = load_golden(name)
        kw = rec["kwargs"]
        x, y = torch.from_numpy(rec["x"])[None], torch.from_numpy(rec["y"])[None]
        L64 = sinkhorn_tensorized_cpu(x, y, p=kw["p"], blur=kw["blur"], scaling=kw.get("scaling", 0.5)).item()
        assert abs(L64 - float(rec["loss_f64"])) <= 1e-9 * abs(float(rec["loss_f64"]))
        L32 = sinkhorn_tensorized_cpu(x.float(), y.float(), p=kw["p"], blur=kw["blur"], scaling=kw.get("scaling", 0.5)).item()
        assert abs(L32 - float(rec["loss_f32"])) <= 2e-5 * abs(float(rec["loss_f32"]))


def test_torch64_two_scale_oracle_equals_the_dense_masked_one():
    """oracle_torch64.sinkhorn_multiscale (fine level cluster by cluster: what runs at N = 1e6 on the test GPU) against
    oracle_np.sinkhorn_multiscale (dense masked matrices) on problems small enough for both."""
    rng = np.random.default_rng(3)
    N, M = 700, 640
    x, y = rng.random((N, 3)), rng.random((M, 3)) * 0.6 + 0.3
    a, b = np.full(N, 1 / N), np.full(M, 1 / M)
    for kw in (dict(scaling=0.7), dict(scaling=0.7, truncate=2), dict(diameter=1.0, cluster_scale=0.1), dict(scaling=0.7, debias=False)):
        (r, g), i1 = oracle_np.sinkhorn_multiscale(a, x, b, y, grad=True, return_info=True, **kw)
        (r2, g2), i2 = oracle_torch64.sinkhorn_multiscale(a, x, b, y, grad=True, return_info=True, device=CPU, **kw)
        assert abs(r - r2) <= 1e-12 * abs(r) and relerr(g2, g) < 1e-11
        assert np.allclose(i1["kept_fraction"], i2["kept_fraction"], rtol=1e-12) and i1["jumps"] == i2["jumps"]
        F, G = oracle_np.sinkhorn_multiscale(a, x, b, y, potentials=True, **kw)
        F2, G2 = oracle_torch64.sinkhorn_multiscale(a, x, b, y, potentials=True, device=CPU, **kw)
        assert np.abs(F - F2).max() < 1e-12 and np.abs(G - G2).max() < 1e-12
        full = oracle_torch64.sinkhorn_multiscale(a, x, b, y, full=True, device=CPU, **kw)      # everything from one run
        assert full["loss"] == r2 and np.array_equal(full["gx"], g2) and np.array_equal(full["F"], F2) and np.array_equal(full["G"], G2)


@pytest.mark.parametrize("budget", [1 << 27, 40_000, 3_000])
@pytest.mark.parametrize("p", [2, 1])
def test_torch64_fine_level_runs_of_clusters(monkeypatch, budget, p):
    """The batched fine level (runs of consecutive row clusters against the union of their kept columns, foreign pairs masked
    out again) reduces over exactly the kept pair set whatever the grouping: one run for everything, a few clusters per run,
    and runs so small that a single cluster is cut into row chunks — all equal to the dense masked oracle."""
    monkeypatch.setattr(oracle_torch64, "_FINE_BUDGET", budget)
    rng = np.random.default_rng(5)
    N, M = 500, 450
    x, y = rng.random((N, 3)), rng.random((M, 3)) * 0.7 + 0.2
    a, b = rng.random(N) + 0.5, rng.random(M) + 0.5
    a, b = a / a.sum(), b / b.sum()
    kw = dict(p=2, blur=0.05, scaling=0.7, truncate=2, cluster_scale=0.15) if p == 2 else dict(p=1, blur=0.02, scaling=0.7, truncate=1, cluster_scale=0.1)
    (r, g), i1 = oracle_np.sinkhorn_multiscale(a, x, b, y, grad=True, return_info=True, **kw)
    full = oracle_torch64.sinkhorn_multiscale(a, x, b, y, full=True, device=CPU, **kw)
    assert 0 < i1["kept_fraction"][0] < 0.9 and np.allclose(i1["kept_fraction"], full["info"]["kept_fraction"], rtol=1e-12)
    assert abs(r - full["loss"]) <= 1e-11 * abs(r) and relerr(full["gx"], g) < 1e-10
    F, G = oracle_np.sinkhorn_multiscale(a, x, b, y, potentials=True, **kw)
    assert np.abs(F - full["F"]).max() < 1e-12 and np.abs(G - full["G"]).max() < 1e-12


def test_hip64_pattern_covers_exactly_the_kept_blocks():
    """``oracle_hip64.make_pattern`` (host side of the float64 HIP oracle: cluster keep mask -> per-row CSR lists of column
    intervals) expands to the same point-level mask as ``oracle_np._expand_mask``, the construction the dense two-scale oracle
    uses — including row clusters that keep nothing."""
    from oracle import oracle_hip64
    rng = np.random.default_rng(4)
    N, M, ci, cj = 230, 190, 9, 7
    cut = lambda n, c: np.r_[0, np.sort(rng.choice(np.arange(1, n), c - 1, replace=False)), n]      # noqa: E731
    bi, bj = cut(N, ci), cut(M, cj)
    ri, rj = np.stack([bi[:-1], bi[1:]], 1), np.stack([bj[:-1], bj[1:]], 1)
    keep = rng.random((ci, cj)) < 0.4
    keep[3, :] = False
    lab, offsets, intervals = (t.numpy() for t in oracle_hip64.make_pattern(keep, ri, rj, "cpu"))
    mask = np.zeros((N, M), bool)
    for i in range(N):
        for q in range(offsets[lab[i]], offsets[lab[i] + 1]):
            assert not mask[i, intervals[q, 0]:intervals[q, 1]].any()      # intervals of a row are disjoint
            mask[i, intervals[q, 0]:intervals[q, 1]] = True
    assert np.array_equal(mask, oracle_np._expand_mask(keep, ri, rj, N, M))
    empty = oracle_hip64.make_pattern(np.zeros((2, 2), bool), ri[:2], rj[:2], "cpu")
    assert empty[1].tolist() == [0, 0, 0] and empty[2].shape == (1, 2)


# ---- the two-scale drivers, pinned to runs of the reference's own code (tests/golden/make_golden_multiscale.py) --------------

from conftest import multiscale_cases  # noqa: E402


def _two_scale_kwargs(rec):
    kw = dict(rec["kwargs"])
    return kw.pop("loss"), kw


@pytest.mark.parametrize("name", multiscale_cases())
def test_two_scale_oracle_matches_reference_driver(name):
    """oracle_np.sinkhorn_multiscale (what the HIP two-scale path is held to, also at N = 1e6 through oracle_torch64) against
    the reference's own `sinkhorn_multiscale` (`_legacy/sinkhorn_samples.py:547-681`) run in float64 with dense stand-ins for
    its pykeops primitives: loss, gradient (balanced cases), potentials in the caller's point order."""
    rec = load_golden(name)
    _, kw = _two_scale_kwargs(rec)
    a, x, b, y = rec["a"], rec["x"], rec["b"], rec["y"]
    L, info = oracle_np.sinkhorn_multiscale(a, x, b, y, return_info=True, **kw)
    assert abs(L - float(rec["loss_f64"])) <= 1e-7 * abs(float(rec["loss_f64"]))
    if name == "multiscale_p2_last_jump":      # cluster scale below the blur: the jump is the last step
        assert info["jumps"][0] == len(info["eps_list"]) - 1
    else:
        assert info["jumps"][0] < len(info["eps_list"]) - 1
    F, G = oracle_np.sinkhorn_multiscale(a, x, b, y, potentials=True, **kw)
    assert relerr(F, rec["F_f64"]) < 1e-7 and relerr(G, rec["G_f64"]) < 1e-7
    if "reach" not in name:
        _, gx = oracle_np.sinkhorn_multiscale(a, x, b, y, grad=True, **kw)
        assert relerr(gx, rec["gx_f64"]) < 1e-7
    # the reference's own float32 run of the same problem is no further than 1e-4 (what the HIP path is held to)
    assert abs(float(rec["loss_f32"]) - float(rec["loss_f64"])) <= 1e-4 * abs(float(rec["loss_f64"]))


@pytest.mark.parametrize("name", [n for n in multiscale_cases() if "reach" not in n])
def test_torch64_two_scale_oracle_matches_reference_driver(name):
    """The chunked float64 oracle of the N = 1e6 tests, on the same reference runs."""
    rec = load_golden(name)
    _, kw = _two_scale_kwargs(rec)
    full = oracle_torch64.sinkhorn_multiscale(rec["a"], rec["x"], rec["b"], rec["y"], full=True, device=CPU, **kw)
    assert abs(full["loss"] - float(rec["loss_f64"])) <= 1e-7 * abs(float(rec["loss_f64"]))
    assert relerr(full["gx"], rec["gx_f64"]) < 1e-7
    assert relerr(full["F"], rec["F_f64"]) < 1e-7 and relerr(full["G"], rec["G_f64"]) < 1e-7


def test_p1_clamp_versus_keops_norm2_is_documented_size():
    """p = 1: the fixtures (and the HIP kernels) follow the tensorized clamp sqrt(max(d2, 1e-8)) (`_legacy/utils.py:56-61`);
    KeOps' `Norm2` has no clamp.  The reference run with the un-clamped root is stored beside: the two differ by 1e-5..1e-3."""
    for name in multiscale_cases():
        rec = load_golden(name)
        if "loss_f64_keops_norm2" in rec:
            d = abs(float(rec["loss_f64_keops_norm2"]) - float(rec["loss_f64"])) / abs(float(rec["loss_f64"]))
            assert 1e-5 < d < 1e-3, (name, d)


def _unsort(v, perm):
    out = np.empty_like(v)
    out[perm] = v
    return out


@pytest.mark.parametrize("name", multiscale_cases(kernels=True))
def test_kernel_two_scale_oracle_matches_reference_driver(name):
    """oracle_np.kernel_multiscale against the reference's own `kernel_multiscale` (`_legacy/kernel_samples.py:177-271`) run in
    float64 with a dense `LazyTensor` carrying KeOps ranges.  The reference returns the potentials in cluster-sorted order."""
    rec = load_golden(name)
    loss, kw = _two_scale_kwargs(rec)
    a, x, b, y = rec["a"], rec["x"], rec["b"], rec["y"]
    L, info = oracle_np.kernel_multiscale(loss, a, x, b, y, return_info=True, **kw)
    assert abs(L - float(np.asarray(rec["loss_f64"]).reshape(-1)[0])) <= 1e-9 * abs(L)
    if loss != "energy":
        assert all(0 < k < 1 for k in info["kept_fraction"]), info["kept_fraction"]
    (F, G), info = oracle_np.kernel_multiscale(loss, a, x, b, y, potentials=True, return_info=True, **kw)
    Fr, Gr = rec["F_f64"].reshape(-1), rec["G_f64"].reshape(-1)
    if "perm_x" in rec:        # both sides return cluster-SORTED potentials (`:232-233`); the order inside a cluster is the sort's
        F, Fr = _unsort(F, info["perm_x"]), _unsort(Fr, rec["perm_x"])
        G, Gr = _unsort(G, info["perm_y"]), _unsort(Gr, rec["perm_y"])
    assert relerr(F, Fr) < 1e-9 and relerr(G, Gr) < 1e-9
    if loss != "energy":
        _, gx = oracle_np.kernel_multiscale(loss, a, x, b, y, grad=True, **kw)
        assert relerr(gx, rec["gx_f64"]) < 1e-8


def test_voxel_bins_follow_the_precision_of_the_cloud():
    """`grid_cluster` evaluates `(x / size).floor()` in the dtype of x (pykeops.torch.cluster.grid_cluster on a float32 cloud works in
    float32).  Among 1e6 float32 points a handful have a quotient on the other side of an integer in float64 — another voxel, another
    coarse problem: ``label_dtype`` makes the oracle bin as a float32 run does (tests/test_full_size_gpu.py relies on it)."""
    rng = np.random.default_rng(0)
    size = 1.0 / 2000 ** (1 / 3)
    x = rng.random(3_000_000).astype(np.float32)
    q32 = np.floor(x / np.float32(size)).astype(np.int64)
    q64 = np.floor(x.astype(np.float64) / size).astype(np.int64)
    moved = np.flatnonzero(q32 != q64)
    assert 0 < moved.size < 50                                     # rare, but there
    pts = np.stack([x[moved[0]] + np.zeros(4, np.float32), np.linspace(0, 1, 4, dtype=np.float32)], 1)
    pts = np.concatenate([pts, np.array([[q64[moved[0]] * size + 0.5 * size, 0.0]], np.float32)])     # a point well inside the float64 voxel
    lab32 = oracle_np.grid_cluster(pts.astype(np.float64), size, label_dtype=np.float32)
    lab64 = oracle_np.grid_cluster(pts.astype(np.float64), size)
    assert lab64[0] == lab64[4] and lab32[0] != lab32[4]
    torch_lab = torch.floor(torch.from_numpy(pts) / size).long()   # what torch does on the float32 tensor
    assert np.array_equal(torch_lab[:, 0].numpy(), np.floor(pts[:, 0] / np.float32(size)).astype(np.int64))
