"""CPU tests of the host side: SamplesLoss mirror (shapes, errors, dispatch), tensorized path vs the
reference's golden outputs, epsilon schedule, clustering / range helpers, C-ABI symbols.  No GPU."""

import ctypes
import os
import re
import warnings

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_cases, load_golden, relerr
from geomloss_amd import SamplesLoss, hip
from geomloss_amd import sinkhorn_divergence as sd
from geomloss_amd.cluster import cluster_ranges_centroids, from_matrix, grid_cluster
from oracle import oracle_np


# ---- tensorized path == reference (fp32 bit-level on cfg 1, 1e-6 elsewhere) --------------------

def test_cfg1_tensorized_cpu_equals_reference_fp32():
    rec = load_golden("cfg1_n2000_d2")
    L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="tensorized")(torch.from_numpy(rec["x"]), torch.from_numpy(rec["y"]))
    assert L.item() == pytest.approx(float(rec["loss_f32"]), rel=1e-6)


@pytest.mark.parametrize("name", golden_cases())
def test_tensorized_cpu_matches_golden(name):
    rec = load_golden(name)
    a, x, b, y = (torch.from_numpy(rec[k]).float() for k in "axby")
    x.requires_grad_(True)
    L = SamplesLoss(backend="tensorized", **rec["kwargs"])(a, x, b, y)
    assert relerr(L.detach().numpy(), rec["loss_f32"]) < 2e-5
    (gx,) = torch.autograd.grad(L.sum(), [x])
    assert relerr(gx.numpy(), rec["gx_f32"]) < 2e-4
    F, G = SamplesLoss(backend="tensorized", potentials=True, **rec["kwargs"])(a, x.detach(), b, y)
    assert F.shape == rec["F_f32"].shape and G.shape == rec["G_f32"].shape  # incl. the (1,N) quirk
    assert relerr(F.numpy(), rec["F_f32"]) < 2e-5


def test_epsilon_schedule_lengths_match_survey_probe():
    # 7 (D=2 unit square), 8 (D=3 unit cube), 7 (diameter=1), 9 (blur=.01), 36 (scaling .9)
    assert len(sd.epsilon_schedule(2, np.sqrt(2), 0.05, 0.5)) == 7
    assert len(sd.epsilon_schedule(2, np.sqrt(3), 0.05, 0.5)) == 8
    assert len(sd.epsilon_schedule(2, 1.0, 0.05, 0.5)) == 7
    assert len(sd.epsilon_schedule(2, 1.0, 0.01, 0.5)) == 9
    assert len(sd.epsilon_schedule(2, np.sqrt(3), 0.05, 0.9)) == 36
    assert sd.epsilon_schedule(2, 1.7, 0.05, 0.5) == oracle_np.epsilon_schedule(2, 1.7, 0.05, 0.5)


def test_log_weights_and_dampening():
    a = torch.tensor([0.5, 0.0, 0.5])
    assert sd.log_weights(a).tolist() == [pytest.approx(np.log(0.5)), -100000.0, pytest.approx(np.log(0.5))]
    assert sd.dampening(0.1, None) == 1 and sd.dampening(0.5, 2.0) == pytest.approx(0.8)
    # gradients go through the forward factor rho + eps/2 (reference quirk)
    w = sd.UnbalancedWeight(0.5, 2.0)
    t = torch.ones(1, requires_grad=True)
    w(t).backward()
    assert t.grad.item() == pytest.approx(2.25)


def test_sinkhorn_loop_with_an_iter4_hook_matches_the_plain_loop():
    """The optional one-launch-per-iteration hook of sinkhorn_loop is semantically the four simultaneous updates."""
    import torch
    from geomloss_amd.sinkhorn_divergence import epsilon_schedule, log_weights, sinkhorn_loop
    from geomloss_amd.sinkhorn_samples import softmin_tensorized
    from geomloss_amd.utils import squared_distances

    torch.manual_seed(0)
    x, y = torch.rand(1, 40, 2), torch.rand(1, 50, 2)
    a, b = torch.full((1, 40), 1 / 40), torch.full((1, 50), 1 / 50)
    C = lambda u, v: squared_distances(u, v) / 2  # noqa: E731
    costs = dict(C_xxs=C(x, x), C_yys=C(y, y), C_xys=C(x, y), C_yxs=C(y, x))
    eps_list = epsilon_schedule(2, 1.5, 0.1, 0.6)

    class Fused:
        calls = 0

        def __call__(self, eps, Cm, h):
            return softmin_tensorized(eps, Cm, h)

        def iter4(self, eps, C_xy, a_log, b_log, pots, damping, debias):
            Fused.calls += 1
            sm = lambda Cm, lw, pot, prev: (damping * softmin_tensorized(eps, Cm, lw if pot is None else lw + pot / eps)  # noqa: E731
                                            if prev is None else
                                            0.5 * (prev + damping * softmin_tensorized(eps, Cm, lw + pot / eps)))
            p4 = (None,) * 4 if pots is None else tuple(pots) + (None,) * (4 - len(pots))
            out = [sm(costs["C_xys"], b_log, p4[1], p4[0]), sm(costs["C_yxs"], a_log, p4[0], p4[1])]
            if debias:
                out += [sm(costs["C_xxs"], a_log, p4[2], p4[2]), sm(costs["C_yys"], b_log, p4[3], p4[3])]
            return tuple(out)

    for debias in (True, False):
        args = (log_weights(a), log_weights(b), costs["C_xxs"] if debias else None, costs["C_yys"] if debias else None,
                costs["C_xys"], costs["C_yxs"], eps_list, 0.7)
        plain = sinkhorn_loop(softmin_tensorized, *args, debias=debias)
        Fused.calls = 0
        fused = sinkhorn_loop(Fused(), *args, debias=debias)
        assert Fused.calls == len(eps_list) + 1
        for u, v in zip(plain, fused):
            assert (u is None and v is None) or torch.allclose(u, v, atol=1e-6)


def test_sinkhorn_loop_with_an_extrapolation_hook_matches_the_plain_two_level_loop():
    """`extrapolate.all4` (the coarse-to-fine jump of every potential as one call) is semantically the four `extrapolate` calls; a
    hook that answers None leaves them in place; and after a jump that follows the LAST iteration (the differentiable step) the
    hook is not asked."""
    from functools import partial

    from geomloss_amd.sinkhorn_divergence import epsilon_schedule, log_weights_many, sinkhorn_loop
    from geomloss_amd.sinkhorn_samples import softmin_tensorized
    from geomloss_amd.utils import squared_distances

    torch.manual_seed(1)
    xc, yc, x, y = torch.rand(1, 7, 2), torch.rand(1, 9, 2), torch.rand(1, 40, 2), torch.rand(1, 50, 2)
    ws = [torch.full((1, n), 1.0 / n) for n in (7, 40, 9, 50)]
    la_c, la, lb_c, lb = log_weights_many(ws)
    C = lambda u, v: squared_distances(u, v) / 2  # noqa: E731
    # cost objects: (matrix, rows, columns) so that an extrapolation can build the fine-rows x coarse-columns matrix
    obj = lambda u, v: (C(u, v), u, v)  # noqa: E731
    softmin = lambda eps, Cm, h: softmin_tensorized(eps, Cm[0], h)  # noqa: E731

    def extrapolate(f_ba, g_ab, eps, damping, C_xy, b_log, C_xy_fine):
        return damping * softmin_tensorized(eps, C(C_xy_fine[1], C_xy[2]), b_log + g_ab / eps)

    calls = []

    def all4(answer, pots, eps, damping, C_xy, C_yx, a_log, b_log, C_xy_fine, C_yx_fine, debias):
        calls.append(torch.is_grad_enabled())
        if not answer:
            return None
        out = [extrapolate(pots[0], pots[1], eps, damping, C_xy, b_log, C_xy_fine), extrapolate(pots[1], pots[0], eps, damping, C_yx, a_log, C_yx_fine)]
        if debias:
            out += [extrapolate(pots[2], pots[2], eps, damping, (None, C_xy[1], C_xy[1]), a_log, C_xy_fine),
                    extrapolate(pots[3], pots[3], eps, damping, (None, C_yx[1], C_yx[1]), b_log, C_yx_fine)]
        return tuple(out)

    eps_list = epsilon_schedule(2, 1.5, 0.1, 0.6)
    for debias in (True, False):
        args = ([la_c, la], [lb_c, lb], [obj(xc, xc), obj(x, x)] if debias else None, [obj(yc, yc), obj(y, y)] if debias else None,
                [obj(xc, yc), obj(x, y)], [obj(yc, xc), obj(y, x)], eps_list, 0.7)
        kw = dict(kernel_truncation=lambda C_xy, C_yx, C_xy_, C_yx_, *a, **k: (C_xy_, C_yx_), debias=debias)
        plain = sinkhorn_loop(softmin, *args, jumps=[3], extrapolate=extrapolate, **kw)
        for answer in (True, False):
            hooked = partial(extrapolate)
            hooked.all4 = partial(all4, answer)
            calls.clear()
            got = sinkhorn_loop(softmin, *args, jumps=[3], extrapolate=hooked, **kw)
            assert calls == [False]          # asked once, with autograd off
            for u, v in zip(plain, got):
                assert (u is None and v is None) or torch.allclose(u, v, atol=1e-6)
        hooked = partial(extrapolate)
        hooked.all4 = partial(all4, True)
        calls.clear()
        sinkhorn_loop(softmin, *args, jumps=[len(eps_list) - 1], extrapolate=hooked, **kw)
        assert calls == []               # the jump after the last iteration is the differentiable step: four plain calls


@pytest.mark.parametrize("jumps", [[], [3]])
def test_sinkhorn_loop_with_an_anneal_hook_matches_the_plain_loop(jumps):
    """`softmin.anneal` (the initialisation and every iteration of the first level in one call) is semantically those iterations:
    a single-scale loop hands it the whole schedule, a two-level one the temperatures up to its jump; None leaves the loop alone."""
    from geomloss_amd.sinkhorn_divergence import epsilon_schedule, log_weights_many, sinkhorn_loop
    from geomloss_amd.sinkhorn_samples import softmin_tensorized
    from geomloss_amd.utils import squared_distances

    torch.manual_seed(2)
    xc, yc, x, y = torch.rand(1, 7, 2), torch.rand(1, 9, 2), torch.rand(1, 40, 2), torch.rand(1, 50, 2)
    la_c, la, lb_c, lb = log_weights_many([torch.full((1, n), 1.0 / n) for n in (7, 40, 9, 50)])
    C = lambda u, v: squared_distances(u, v) / 2  # noqa: E731
    obj = lambda u, v: (C(u, v), u, v)  # noqa: E731
    eps_list = epsilon_schedule(2, 1.5, 0.1, 0.6)

    def extrapolate(f_ba, g_ab, eps, damping, C_xy, b_log, C_xy_fine):
        return damping * softmin_tensorized(eps, C(C_xy_fine[1], C_xy[2]), b_log + g_ab / eps)

    class Soft:
        seen = None

        def __init__(self, answer):
            self.answer = answer

        def __call__(self, eps, Cm, h):
            return softmin_tensorized(eps, Cm[0], h)

        def anneal(self, eps_l, dampings, C_xy, a_log, b_log, debias):
            Soft.seen = (list(eps_l), list(dampings))
            if not self.answer:
                return None
            xs, ys = C_xy[1], C_xy[2]
            costs = [(C(xs, ys), b_log), (C(ys, xs), a_log)] + ([(C(xs, xs), a_log), (C(ys, ys), b_log)] if debias else [])
            src = [1, 0, 2, 3]      # the potential each update reads: g_ab for f_ba, f_ba for g_ab, its own for the debiasing pair
            pots = [dampings[0] * softmin_tensorized(eps_l[0], Cm, lw) for Cm, lw in costs]
            old = pots
            for e, d in zip(eps_l, dampings):
                old = pots
                pots = [0.5 * (old[k] + d * softmin_tensorized(e, costs[k][0], costs[k][1] + old[src[k]] / e)) for k in range(len(costs))]
            return tuple(pots), tuple(old)

    for debias in (True, False):
        if jumps:
            args = ([la_c, la], [lb_c, lb], [obj(xc, xc), obj(x, x)] if debias else None, [obj(yc, yc), obj(y, y)] if debias else None,
                    [obj(xc, yc), obj(x, y)], [obj(yc, xc), obj(y, x)], eps_list, 0.7)
            kw = dict(jumps=jumps, extrapolate=extrapolate, kernel_truncation=lambda C_xy, C_yx, C_xy_, C_yx_, *a, **k: (C_xy_, C_yx_))
        else:
            args = (la, lb, obj(x, x) if debias else None, obj(y, y) if debias else None, obj(x, y), obj(y, x), eps_list, 0.7)
            kw = {}
        plain = sinkhorn_loop(lambda eps, Cm, h: softmin_tensorized(eps, Cm[0], h), *args, debias=debias, **kw)
        for answer in (True, False):
            got = sinkhorn_loop(Soft(answer), *args, debias=debias, **kw)
            want = eps_list[: jumps[0] + 1] if jumps else eps_list
            assert Soft.seen[0] == list(want) and Soft.seen[1] == [sd.dampening(e, 0.7) for e in want]
            for u, v in zip(plain, got):
                assert (u is None and v is None) or torch.allclose(u, v, atol=1e-6)


def test_log_weights_many_and_uniform_weights_are_the_one_by_one_values():
    from geomloss_amd.samples_loss import _uniform_weight
    ws = [torch.tensor([0.5, 0.0, -1.0, 1e-45, 3.0]), torch.rand(7), torch.zeros(3)]
    for got, w in zip(sd.log_weights_many(ws), ws):
        ref = w.log()
        ref[w <= 0] = -100000.0       # the reference's masked assignment (sinkhorn_divergence.py:61-65)
        assert torch.equal(got, ref) and torch.equal(sd.log_weights(w), ref)
    for dt in (torch.float32, torch.bfloat16, torch.float16, torch.float64):
        for N in (3, 7, 1000, 999983, 16777217):
            assert torch.equal(torch.full((4,), _uniform_weight(N, dt), dtype=dt), torch.ones(4, dtype=dt) / N)
    x = torch.rand(2, 5, 3, dtype=torch.float64)
    w = SamplesLoss("sinkhorn", backend="tensorized").generate_weights(x)
    assert w.shape == (2, 5) and w.dtype == torch.float64 and torch.equal(w, torch.ones(2, 5, dtype=torch.float64) / 5)


def test_sinkhorn_loop_leaves_grad_enabled():
    x, y = torch.rand(20, 2), torch.rand(30, 2)
    with torch.no_grad():
        SamplesLoss("sinkhorn", backend="tensorized")(x, y)
        assert torch.is_grad_enabled()  # reference behaviour: sinkhorn_divergence.py:434,612


# ---- shapes, call forms, error messages ---------------------------------------------------------

def test_call_forms_and_output_shapes():
    L = SamplesLoss("gaussian", blur=0.5, backend="tensorized")
    x, y = torch.rand(10, 3), torch.rand(12, 3)
    a, b = torch.full((10,), 0.1), torch.full((12,), 1 / 12)
    assert L(x, y).shape == () and L(a, x, b, y).shape == ()
    assert L(a[:, None], x, b[:, None], y).item() == pytest.approx(L(a, x, b, y).item())
    xb, yb = torch.rand(4, 10, 3), torch.rand(4, 12, 3)
    assert L(xb, yb).shape == (4,)
    assert L(a.expand(4, 10)[..., None], xb, b.expand(4, 12)[..., None], yb).shape == (4,)
    F, G = SamplesLoss("sinkhorn", potentials=True, backend="tensorized")(x, y)
    assert F.shape == (1, 10) and G.shape == (1, 12)  # reference quirk: view_as the unsqueezed weights
    with pytest.raises(ValueError, match="two .x, y., four"):
        L(x, y, x)


@pytest.mark.parametrize("args,msg", [
    (lambda: (torch.rand(5), torch.rand(5, 2), torch.rand(6, 1), torch.rand(6, 2)), "same number of dimensions"),
    (lambda: (torch.rand(5), torch.rand(5, 2), torch.rand(6), torch.rand(1, 6, 2)), "samples 'x' and 'y' should have the same number"),
    (lambda: (torch.rand(5), torch.rand(5, 2), torch.rand(6), torch.rand(6, 3)), "same last dimension"),
    (lambda: (torch.rand(5, 2), torch.rand(5, 2), torch.rand(6, 2), torch.rand(6, 2)), r"'α' should be encoded as \(N,\) or \(N,1\)"),
    (lambda: (torch.rand(4), torch.rand(5, 2), torch.rand(6), torch.rand(6, 2)), "Weights 'α' and samples 'x'"),
    (lambda: (torch.rand(5), torch.rand(5, 2), torch.rand(7), torch.rand(6, 2)), "Weights 'β' and samples 'y'"),
    (lambda: (torch.rand(2, 5), torch.rand(2, 5, 2), torch.rand(3, 6), torch.rand(3, 6, 2)), "same batchsize"),
    (lambda: (torch.rand(5), torch.rand(5), torch.rand(5), torch.rand(5)), r"\(N,D\) or \(B,N,D\)"),
])
def test_shape_errors(args, msg):
    with pytest.raises(ValueError, match=msg):
        SamplesLoss("energy", backend="tensorized")(*args())


def test_labels_and_backend_rules():
    x, y = torch.rand(5, 2), torch.rand(6, 2)
    a, b = torch.full((5,), 0.2), torch.full((6,), 1 / 6)
    with pytest.raises(ValueError, match="Explicit cluster labels"):
        SamplesLoss("sinkhorn", backend="online")(torch.zeros(5).int(), a, x, torch.zeros(6).int(), b, y)
    with pytest.raises(ValueError, match="labels 'l_x' should have the same length"):
        SamplesLoss("sinkhorn", backend="multiscale")(torch.zeros(4).int(), a, x, None, b, y)
    with pytest.raises(NotImplementedError, match="not been implemented with batches"):
        SamplesLoss("sinkhorn")(torch.zeros(1, 5), a[None], x[None], None, b[None], y[None])
    with pytest.raises(KeyError):  # reference: "hausdorff" has no kernel name -> KeyError(None)
        SamplesLoss("hausdorff", backend="tensorized")(x, y)


def test_auto_backend_heuristic():
    L = SamplesLoss("sinkhorn")
    assert L._choose_backend(None, None, 0, 5000, 5000, 3) == "tensorized"
    assert L._choose_backend(None, None, 0, 5001, 5000, 3) == "online"
    assert L._choose_backend(None, None, 0, 10001, 10000, 3) == "multiscale"
    assert L._choose_backend(None, None, 0, 10001, 10000, 4) == "online"
    assert SamplesLoss("sinkhorn", p=1)._choose_backend(None, None, 0, 20000, 20000, 3) == "online"
    assert SamplesLoss("gaussian")._choose_backend(None, None, 0, 20000, 20000, 3) == "online"
    # CPU tensors keep the reference's choice for small clouds
    import torch
    assert L._choose_backend(None, None, 0, 2000, 2000, 3, torch.zeros(2000, 3)) == "tensorized"


def test_multiscale_with_batch_warns_and_falls_back_to_tensorized():
    xb, yb = torch.rand(2, 30, 2), torch.rand(2, 40, 2)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = SamplesLoss("sinkhorn", backend="multiscale")(xb, yb)
    assert out.shape == (2,) and any("do not support batchsize" in str(m.message) for m in w)


def test_hip_backends_refuse_cpu_tensors_loudly():
    x, y = torch.rand(50, 3), torch.rand(60, 3)
    for backend in ("online", "multiscale"):
        with pytest.raises(RuntimeError, match="need GPU tensors|HIP extension"):
            SamplesLoss("sinkhorn", backend=backend)(x, y)
    with pytest.raises(RuntimeError, match="need GPU tensors|HIP extension"):
        SamplesLoss("gaussian", backend="online")(x, y)
    with pytest.raises(NotImplementedError, match="cost formulas"):
        SamplesLoss("sinkhorn", backend="online", cost="Exp(X-Y)")(x, y)
    # the refusal is raised inside the hand-disabled autograd region of the loop: it must not leave gradients off
    assert torch.is_grad_enabled()
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            SamplesLoss("sinkhorn", backend="online")(x, y)
        assert not torch.is_grad_enabled()


# ---- clustering and block-sparse ranges ----------------------------------------------------------

def test_grid_cluster_and_centroids_match_oracle():
    rng = np.random.default_rng(0)
    x = rng.random((500, 3)).astype(np.float32)
    a = rng.random(500).astype(np.float32)
    a /= a.sum()
    lab = grid_cluster(torch.from_numpy(x), 0.3).numpy()
    assert (lab == oracle_np.grid_cluster(x, np.float32(0.3))).all()
    ranges, xc, ac = cluster_ranges_centroids(torch.from_numpy(x), torch.from_numpy(lab), torch.from_numpy(a))
    a_c, _, x_c, _, rng_o, _ = oracle_np.clusterize(a.astype(np.float64), x.astype(np.float64), 0.3)
    assert (ranges.numpy() == rng_o).all()
    assert relerr(xc.numpy(), x_c) < 1e-5 and relerr(ac.numpy(), a_c) < 1e-5


def test_from_matrix_covers_exactly_the_kept_pairs_in_both_orientations():
    rng = np.random.default_rng(1)
    ci, cj = 7, 9
    si, sj = rng.integers(1, 6, ci), rng.integers(1, 6, cj)
    ri = torch.tensor(np.stack([np.cumsum(si) - si, np.cumsum(si)], 1), dtype=torch.int32)
    rj = torch.tensor(np.stack([np.cumsum(sj) - sj, np.cumsum(sj)], 1), dtype=torch.int32)
    keep = torch.from_numpy(rng.random((ci, cj)) < 0.5)
    keep[2] = False
    rg = from_matrix(ri, rj, keep)

    def covered(ranges_i, slices_i, red_j, n_i, n_j):
        m = np.zeros((n_i, n_j), int)
        for k in range(len(ranges_i)):
            q0 = 0 if k == 0 else int(slices_i[k - 1])
            for q in range(q0, int(slices_i[k])):
                m[ranges_i[k, 0]:ranges_i[k, 1], red_j[q, 0]:red_j[q, 1]] += 1
        return m

    N, M = int(si.sum()), int(sj.sum())
    want = oracle_np._expand_mask(keep.numpy(), ri.numpy(), rj.numpy(), N, M).astype(int)
    got = covered(rg.ranges_i.numpy(), rg.slices_i.numpy(), rg.redranges_j.numpy(), N, M)
    assert (got == want).all()  # every kept pair exactly once, nothing else
    got_t = covered(rg.ranges_j.numpy(), rg.slices_j.numpy(), rg.redranges_i.numpy(), M, N)
    assert (got_t == want.T).all()
    # adjacent kept clusters are merged into one interval
    n_runs = sum(len(re.findall("1+", "".join("1" if v else "0" for v in row))) for row in keep.numpy())
    assert rg.redranges_j.shape[0] == n_runs


def test_kernel_norm_runs_on_the_union_cloud(monkeypatch):
    """The matrix-free dense kernel norm is 1/2 <w, K_zz w> on the union cloud z = (x, y) with signed weights w = (a, -b): two
    launches (rows of x, rows of y) whose columns are ALL the points, so that the positive and the negative columns of a row are
    summed by one launch around one centre (kernel_samples._kernel_loss_union); default kernel family whatever the gradients.
    The block-sparse (multiscale) norm keeps the three products of the reference, in one family when gradients are on
    (GLHIP_FLAG_GRAD_FAMILY).  Host logic only: the launches are recorded, not run."""
    from geomloss_amd import kernel_samples as ks

    header = open(os.path.join(ROOT, "include", "glhip.h")).read()
    assert re.search(r"#define GLHIP_FLAG_GRAD_FAMILY GLHIP_FLAG_XDL16", header) and hip.FLAG_GRAD_FAMILY == hip.FLAG_XDL16
    seen = []

    def record(kind, x, y, v, blur=0.05, ranges=None, flags=0):
        seen.append((kind, int(flags), x.requires_grad, tuple(x.shape), tuple(y.shape), v.detach().clone()))
        return (v.sum(-1, keepdim=True) + 0 * x.sum(-1)).expand(x.shape[:-1]) if v.dim() == x.dim() - 1 else v

    units = []

    def record_with_unit(kind, x, y, v, blur, want_unit, flags=0):      # the one-node path taken when a gradient is wanted
        units.append(bool(want_unit))
        return record(kind, x, y, v, blur, None, flags), None

    monkeypatch.setattr(hip, "kernel_conv", record)
    monkeypatch.setattr(hip, "kernel_conv_with_unit", record_with_unit)
    x, y = torch.rand(7, 3), torch.rand(5, 3)
    a, b = torch.full((7,), 1 / 7), torch.full((5,), 1 / 5)
    for name in ("gaussian", "laplacian", "energy"):
        for grad in (False, True):
            seen.clear()
            units.clear()
            out = ks.kernel_online(a, x.clone().requires_grad_(grad), b, y, blur=0.1, name=name)
            assert [(s[3], s[4]) for s in seen] == [((7, 3), (12, 3)), ((5, 3), (12, 3))] and {s[1] for s in seen} == {0}
            assert units == ([True, False] if grad else []) and out.requires_grad == grad     # product + gradient in one pass for x only
            for s in seen:
                assert torch.equal(s[5], torch.cat((a, -b)))
    seen.clear()                  # batches: the union along the point axis
    ks.kernel_online(a.expand(2, -1), x.expand(2, -1, -1), b.expand(2, -1), y.expand(2, -1, -1), blur=0.1, name="gaussian")
    assert [(s[3], s[4]) for s in seen] == [((2, 7, 3), (2, 12, 3)), ((2, 5, 3), (2, 12, 3))]
    # block-sparse norm: three products, one family under gradients
    R = object()
    for grad, want in ((False, {0}), (True, {hip.FLAG_GRAD_FAMILY})):
        seen.clear()
        ks.kernel_loss(a, x.clone().requires_grad_(grad), b, y, blur=0.1, name="gaussian", use_keops=True, ranges_xx=R, ranges_yy=R, ranges_xy=R)
        assert len(seen) == 3 and {s[1] for s in seen} == want
    seen.clear()
    with torch.no_grad():     # inference on leaves that require gradients: default kernels
        ks.kernel_loss(a, x.clone().requires_grad_(True), b, y, blur=0.1, name="gaussian", use_keops=True, ranges_xx=R, ranges_yy=R, ranges_xy=R)
    assert {s[1] for s in seen} == {0}


def test_upper_triangle_patterns_cover_each_unordered_block_pair_once():
    """kernel_samples._upper_triangle_patterns: the diagonal pattern holds exactly the pairs of a 256-row block with itself, the
    upper pattern exactly the pairs (i, j) with j in a LATER block — together every unordered pair of rows once (diagonal blocks:
    both orientations), which is what `a . (d + 2 u)` needs.  Pure index logic: checked on the CPU with a dense mask."""
    from geomloss_amd import kernel_samples as ks

    for N in (1, 77, 256, 257, 1000, 5000):
        diag, upper = ks._upper_triangle_patterns(N, torch.device("cpu"))
        blk = torch.arange(N) // ks._UPPER_BLOCK
        for pat, want in ((diag, blk[:, None] == blk[None, :]), (upper, blk[None, :] > blk[:, None])):
            mask = torch.zeros(N, N, dtype=torch.int32)
            first = 0
            assert pat.ranges_i.shape[0] == pat.slices_i.shape[0] == (N + ks._UPPER_BLOCK - 1) // ks._UPPER_BLOCK
            for (i0, i1), last in zip(pat.ranges_i.tolist(), pat.slices_i.tolist()):
                for j0, j1 in pat.redranges_j[first:last].tolist():
                    assert 0 <= j0 <= j1 <= N and (j0 == j1 or j0 % 32 == 0 or j0 == i1)
                    mask[i0:i1, j0:j1] += 1
                first = last
            assert first == pat.redranges_j.shape[0] and torch.equal(mask, want.int()), N


def test_kernel_norm_value_only_launches(monkeypatch):
    """No gradient anywhere + a big problem: the quadratic form <w, K_zz w> over the upper triangle of the union cloud in compact
    order — (diagonal, upper) block-sparse launches, float64 combination; any gradient, potentials, batches or small clouds: the
    two row passes; user ranges: the three products of the reference.  Launches recorded, not run."""
    from geomloss_amd import kernel_samples as ks

    seen = []

    def record(kind, x, y, v, blur=0.05, ranges=None, flags=0):
        seen.append((kind, ranges is not None, int(flags), x.shape[-2], y.shape[-2]))
        return torch.ones(x.shape[:-1], dtype=torch.float32) * v.abs().sum() + 0 * x.sum(-1)

    sorted_clouds = []
    monkeypatch.setattr(hip, "kernel_conv", record)
    monkeypatch.setattr(hip, "kernel_conv_with_unit", lambda kind, x, y, v, blur, want_unit, flags=0: (record(kind, x, y, v, blur, None, flags), None))
    monkeypatch.setattr(hip, "compact_order", lambda pts: (sorted_clouds.append(pts.shape[0]), (torch.arange(pts.shape[0]), pts))[1])
    monkeypatch.setattr(ks, "_UPPER_MIN_PAIRS", 0.0)
    x, y = torch.rand(300, 3), torch.rand(280, 3)
    a, b = torch.full((300,), 1 / 300), torch.full((280,), 1 / 280)
    for name in ("gaussian", "laplacian", "energy"):
        seen.clear()
        sorted_clouds.clear()
        L = ks.kernel_online(a, x, b, y, blur=0.1, name=name)
        assert [s[1:] for s in seen] == [(True, 0, 580, 580)] * 2 and sorted_clouds == [580]
        # every stand-in product is |w|_1 = 2, w . (d + 2 u) = (1 - 1) * 6 = 0
        assert L.dtype == torch.float32 and L.shape == () and abs(L.item()) < 1e-6
        assert ks.kernel_online(a[None], x[None], b[None], y[None], blur=0.1, name=name).shape == (1,)
        seen.clear()
        F, G = ks.kernel_online(a, x, b, y, blur=0.1, name=name, potentials=True)
        assert [s[1:] for s in seen] == [(False, 0, 300, 580), (False, 0, 280, 580)] and F.shape == (300,) and G.shape == (280,)
        seen.clear()
        try:
            ks.kernel_online(a, x, b, y, blur=0.1, name=name, ranges_xy=object())
        except Exception:
            pass
        assert [s[3:] for s in seen[:1]] == [(300, 300)]                    # the reference's three products: K_xx first
        for args in ((a, x.clone().requires_grad_(True), b, y), (a, x, b.clone().requires_grad_(True), y)):
            seen.clear()
            out = ks.kernel_online(*args, blur=0.1, name=name)
            assert [s[1:] for s in seen] == [(False, 0, 300, 580), (False, 0, 280, 580)] and out.requires_grad
        seen.clear()
        ks.kernel_online(a.expand(2, -1), x.expand(2, -1, -1), b.expand(2, -1), y.expand(2, -1, -1), blur=0.1, name=name)
        assert len(seen) == 2 and not any(s[1] for s in seen)
    monkeypatch.setattr(ks, "_UPPER_MIN_PAIRS", 2e9)
    seen.clear()
    ks.kernel_online(a, x, b, y, blur=0.1, name="gaussian")
    assert len(seen) == 2 and not any(s[1] for s in seen)


def test_loss_formulas_sum_in_float64_on_the_gpu_only():
    """utils.scal / scal_sum: the reference's float32 expressions on CPU tensors (the tensorized backend is bit-identical to the
    reference there), float64 accumulation behind the `is_cuda` test."""
    from geomloss_amd import utils
    g = torch.Generator().manual_seed(0)
    a, f, b, h = (torch.rand(1000, generator=g) for _ in range(4))
    assert torch.equal(utils.scal(a, f), torch.dot(a, f))
    assert torch.equal(utils.scal_sum(a, f, b, h), torch.dot(a, f) + torch.dot(b, h))
    A, F = a.view(4, 250), f.view(4, 250)
    assert torch.equal(utils.scal(A, F, batch=True), (A * F).sum(1))
    assert not utils._widen(a, f) and not utils._widen(a.double(), f)


def test_fused_half_step_dispatch_follows_the_flags(monkeypatch):
    """hip.fused_step_applies: glhip_sinkhorn_step has a kernel for D <= 3 whatever the flags, for 4 <= D <= 16 only on the default
    p = 2 matrix-core kernel — under GEOMLOSS_HIP_FLAGS = NO_MFMA / DIRECT (README knobs) those dimensions must take the unfused
    composition (round-3 regression: they raised NotImplementedError).  The Sinkhorn drivers ask this function."""
    for env in (0, hip.FLAG_NO_SPLIT, hip.FLAG_XDL16):
        monkeypatch.setattr(hip, "ENV_FLAGS", env)
        assert all(hip.fused_step_applies(D, p) for D in (1, 2, 3) for p in (1, 2))
        assert all(hip.fused_step_applies(D, 2) for D in (4, 5, 8, 16)) and not hip.fused_step_applies(17, 2)
        assert all(hip.fused_step_applies(D, 1) for D in (4, 8, 16)) and not hip.fused_step_applies(17, 1)      # dense p = 1: glhip_dist_xd.h
        assert not any(hip.fused_step_applies(D, 1, 0, True) for D in (4, 8, 16, 17))                            # block-sparse p = 1: composed
        assert all(hip.fused_step_applies(D, 2, 0, True) for D in (4, 8, 16))
    for env in (hip.FLAG_NO_MFMA, hip.FLAG_DIRECT, hip.FLAG_NO_MFMA | hip.FLAG_DIRECT):
        monkeypatch.setattr(hip, "ENV_FLAGS", env)
        assert all(hip.fused_step_applies(D, p) for D in (1, 2, 3) for p in (1, 2))
        assert not any(hip.fused_step_applies(D, p) for D in (4, 5, 8, 16, 17) for p in (1, 2))
    monkeypatch.setattr(hip, "ENV_FLAGS", 0)
    assert not hip.fused_step_applies(5, 2, flags=hip.FLAG_NO_MFMA) and hip.fused_step_applies(3, 2, flags=hip.FLAG_NO_MFMA)

    # the one-launch iteration (glhip_sinkhorn_iter4) runs the default kernel only: any kernel-selection flag switches it off
    from geomloss_amd import sinkhorn_samples as ss0
    monkeypatch.setattr(hip, "Iter4Plan", lambda *a, **k: "plan")
    x3, al = torch.rand(7, 3), torch.zeros(7)
    for env, want in ((0, "plan"), (hip.FLAG_NO_SPLIT, "plan"), (hip.FLAG_NO_MFMA, None), (hip.FLAG_DIRECT, None), (hip.FLAG_XDL16, None),
                      (hip.FLAG_F32_MFMA, None)):
        monkeypatch.setattr(hip, "ENV_FLAGS", env)
        assert ss0._HipSoftmin(2, multiscale=False)._iter4_plan((x3, x3), al, al, True, create=True) == want, env
    monkeypatch.setattr(hip, "ENV_FLAGS", 0)
    # the drivers: _HipSoftmin.step and ot._averaged compose soft-min + arithmetic instead of calling the fused entry point
    from geomloss_amd import sinkhorn_samples as ss
    from geomloss_amd.ot import sinkhorn_ot
    calls = []
    monkeypatch.setattr(hip, "sinkhorn_step", lambda *a, **k: (calls.append("step"), torch.zeros(7))[1])
    monkeypatch.setattr(hip, "softmin", lambda eps, x, y, h, **k: (calls.append("softmin"), torch.zeros(x.shape[0]))[1])
    x, y, lw = torch.rand(7, 5), torch.rand(6, 5), torch.zeros(6)
    for env, want in ((0, ["step"]), (hip.FLAG_NO_MFMA, ["softmin"]), (hip.FLAG_DIRECT, ["softmin"])):
        monkeypatch.setattr(hip, "ENV_FLAGS", env)
        calls.clear()
        ss._HipSoftmin(2, multiscale=False).step(0.1, (x, y), lw.view(1, -1), lw.view(1, -1), 1.0, torch.zeros(1, 7))
        assert calls == want, (env, calls)
        calls.clear()
        sinkhorn_ot._averaged(0.1, 1.0, x, y, lw, lw, torch.zeros(7))
        assert calls == want, (env, calls)


# ---- C-ABI ------------------------------------------------------------------------------------------

def test_shared_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "glhip.h")).read()
    declared = set(re.findall(r"\b(glhip_\w+)\s*\(", header))
    assert declared == set(hip.SIGNATURES), "include/glhip.h and geomloss_amd/hip.py disagree"
    assert hip.library_available(), "libgeomloss_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} is not exported"
    lib.glhip_version.restype = ctypes.c_int
    assert lib.glhip_version() == int(re.search(r"#define GLHIP_VERSION (\d+)", header).group(1))


def test_ctypes_signatures_follow_the_header_prototypes():
    """Every prototype of include/glhip.h against the ctypes signature geomloss_amd/hip.py binds it with: the same number of
    parameters, and the same kind in every position (pointer / int / long / float / double / size_t) — an argument short or a
    float where the header says int would be read from the wrong register without any error."""
    header = open(os.path.join(ROOT, "include", "glhip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = re.findall(r"\b(int|size_t|const char\s*\*)\s+(glhip_\w+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)
    assert {name for _, name, _ in protos} == set(hip.SIGNATURES)
    # (long and long long are one 64-bit kind here, as they are one ctypes class on LP64)
    kinds = {ctypes.c_void_p: "ptr", ctypes.c_int: "int", ctypes.c_long: "long", ctypes.c_longlong: "long", ctypes.c_float: "float",
             ctypes.c_double: "double", ctypes.c_size_t: "size_t", ctypes.c_char_p: "ptr"}

    def kind_of(param):
        param = " ".join(param.split())
        if "*" in param:
            return "ptr"
        for word, k in (("size_t", "size_t"), ("long", "long"), ("double", "double"), ("float", "float"), ("int", "int")):
            if re.search(r"\b" + word + r"\b", param):
                return k
        raise AssertionError(f"unrecognised parameter '{param}'")

    for ret, name, params in protos:
        params = params.strip()
        want = [] if params in ("", "void") else [kind_of(q) for q in params.split(",")]
        restype, argtypes = hip.SIGNATURES[name]
        got = [kinds[t] for t in argtypes]
        assert got == want, f"{name}: header {want} vs ctypes {got}"
        assert kinds[restype] == ("ptr" if "char" in ret else ret.strip()), name


def test_oracle_is_not_imported_by_the_product():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "geomloss_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"


def test_dense_switch_cost_model_and_exponent_gate():
    """sinkhorn_samples.dense_is_cheaper / _goes_dense: the fine level of a two-scale loss stays dense when the pattern keeps most of
    the matrix or its clusters fill a fraction of the 32-row tiles — and only where the dropped pairs cannot matter."""
    import geomloss_amd.sinkhorn_samples as ss
    # measured on an MI355X (profiles/r04_dense_switch.txt): D = 3 keeps 21 % at every N; dense wins up to 3e4 points, the pattern from 1e5
    assert ss.dense_is_cheaper(0.21 * 1e4 ** 2, 10_000, 10_000, 2000, 2000)
    assert ss.dense_is_cheaper(0.21 * 3e4 ** 2, 30_000, 30_000, 2140, 2140)
    assert not ss.dense_is_cheaper(0.21 * 1e5 ** 2, 100_000, 100_000, 2170, 2170)
    assert not ss.dense_is_cheaper(0.21 * 1e6 ** 2, 1_000_000, 1_000_000, 2197, 2197)
    assert ss.dense_is_cheaper(0.88 * 2e5 ** 2, 200_000, 200_000, 2197, 2197)           # 4-D clouds clustered on 3 coordinates
    assert not ss.dense_is_cheaper(0, 0, 10, 0, 3)

    def never():
        raise AssertionError("the pairs are only counted where the switch may apply")

    eps_jump, eps_last = 0.108 ** 2, 0.05 ** 2
    old = ss._DENSE_SWITCH
    try:
        ss.set_dense_switch("1")
        no = (False, False, False)
        assert ss._goes_dense(5, 0.108, 0.05, 10_000, 10_000, 2000, 2000, never) == no           # p = 1: 5 x 2.2 < 16
        assert ss._goes_dense(2, eps_jump, eps_last, 10_000, 10_000, 2000, 2000, never) == no     # truncate = 2
        assert ss._goes_dense(5, eps_jump, None, 10_000, 10_000, 2000, 2000, never) == no         # a caller that does not say
        assert ss._goes_dense(5, eps_jump, eps_last, 10_000, 10_000, 2000, 2000, lambda: (0.21e8, 5e4, 5e4))[0]
        # 1e5 points: the pattern stays; uniform clusters of 46 points take the small-row-block launch, a cloud whose pairs sit in
        # clusters of hundreds does not, whatever the mean cluster is
        assert ss._goes_dense(5, eps_jump, eps_last, 100_000, 100_000, 2170, 2170, lambda: (0.21e10, 47e5, 48e5)) == (False, True, True)
        assert ss._goes_dense(5, eps_jump, eps_last, 100_000, 100_000, 2170, 2170, lambda: (0.21e10, 300e5, 60e5)) == (False, False, True)
        ss.set_dense_switch("0")
        assert ss._goes_dense(5, eps_jump, eps_last, 10_000, 10_000, 2000, 2000, never) == no
        ss.set_dense_switch("always")
        assert ss._goes_dense(2, eps_jump, None, 10, 10, 2, 2, never) == (True, False, False)
    finally:
        ss.set_dense_switch(old)
