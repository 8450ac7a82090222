"""``geomloss_amd.ot.solve_sample`` on the HIP kernels: against outputs of the reference's own solver (tests/golden/ot_*.npz),
against the closed forms the reference's test-suite uses (``/root/reference/tests/test_ot_solve_sample.py`` with
``generators/diracs.py:75-147``: one point on each side; ``generators/permutations.py``: matched points), and the
matrix-free plan operators against the dense plan."""

import numpy as np
import pytest
import torch
from hypothesis import given, settings
from hypothesis import strategies as st
from hypothesis.extra.numpy import arrays as st_arrays

from conftest import load_golden, ot_golden_cases, relerr
from geomloss_amd import ot
from oracle import oracle_ot

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ot_golden_cases())
def test_solve_sample_matches_reference(cuda, name):
    rec = load_golden(name)
    res = ot.solve_sample(rec["x"], rec["y"], a=rec.get("a"), b=rec.get("b"), **rec["kwargs"])
    assert isinstance(res.value, np.ndarray) and res.value.shape == () and res.value.dtype == np.float64
    assert abs(float(res.value) - float(rec["value"])) <= 1e-4 * abs(float(rec["value"]))
    scale = max(np.abs(rec["potential_a"]).max(), np.abs(rec["potential_b"]).max())
    for k in ("potential_a", "potential_b", "potential_aa", "potential_bb"):
        if k in rec:
            got = getattr(res, k)
            assert got.shape == rec[k].shape and got.dtype == np.float64
            assert np.abs(got - rec[k]).max() <= 1e-4 * scale, k
    for k in ("marginal_a", "marginal_b"):
        assert relerr(getattr(res, k), rec[k]) < 1e-4, k
    if "plan" in rec:
        assert relerr(res.plan, rec["plan"]) < 1e-4
    else:
        assert relerr(res.plan[:5], rec["plan_rows"]) < 1e-4


@given(D=st.integers(1, 5), data=st.data(), max_iter=st.integers(1, 50), reg=st.floats(1e-2, 10.0),
       library=st.sampled_from(["numpy", "torch"]), dtype=st.sampled_from(["float32", "float64"]),
       device=st.sampled_from(["cpu", "cuda"]), weights=st.booleans())
@settings(deadline=None, max_examples=40)
def test_correct_values_diracs(D, data, max_iter, reg, library, dtype, device, weights):
    """One source point, one target point: value = |x - y|^2, plan = [[1]], potentials = value / 2 each, for any temperature
    and any number of iterations; results come back in the caller's library, dtype and device."""
    pts = st_arrays(np.float64, (1, D), elements=st.floats(-10, 10))
    x, y = data.draw(pts), data.draw(pts)
    a = np.ones(1) if weights else None

    def cast(v):
        if v is None:
            return None
        v = v.astype(dtype)
        return torch.from_numpy(v).to(device) if library == "torch" else v

    res = ot.solve_sample(cast(x), cast(y), a=cast(a), b=cast(a), reg=reg, max_iter=max_iter)
    C = float(((x - y) ** 2).sum())

    def check(got, want, shape):
        if library == "torch":
            assert isinstance(got, torch.Tensor) and got.device.type == device and str(got.dtype) == "torch." + dtype
            got = got.cpu().numpy()
        else:
            assert isinstance(got, np.ndarray) and str(got.dtype) == dtype
        assert got.shape == shape
        assert np.allclose(got, want, atol=1e-2)

    check(res.value, C, ())
    check(res.plan, 1.0, (1, 1))
    check(res.potential_a + res.potential_b[0], C, (1,))
    check(res.potential_a - res.potential_b, 0.0, (1,))


def test_reference_suite_dirac_cases(cuda):
    """The reference's own ``tests/test_ot_solve_sample.py::test_correct_values_diracs`` cannot run here (no reference tree on the
    GPU box): its 60 drawn cases travel instead (tests/golden/make_golden_ot_diracs.py: the reference's hypothesis strategy,
    derandomized, with library / dtype / weights as drawn) and are replayed through ``geomloss_amd.ot.solve_sample`` under the
    reference's own acceptance rule (tests/check_ot_result.py: atol = 1e-2, potentials up to the usual constant), plus agreement
    with what the reference's solver returned on them."""
    from conftest import reference_accepts, reference_dirac_cases
    cases = reference_dirac_cases()
    assert len(cases) == 60
    for c in cases:
        lib, dtype = str(c["library"]), str(c["dtype"])

        def cast(v):
            if v is None:
                return None
            v = np.asarray(v).astype(dtype)
            return torch.from_numpy(v).to(cuda) if lib == "torch" else v
        res = ot.solve_sample(cast(c["X_a"]), cast(c["X_b"]), a=cast(c.get("a")), b=cast(c.get("b")), reg=float(c["reg"]),
                              max_iter=int(c["max_iter"]), method=str(c["method"]))
        got = {}
        for k in ("value", "plan", "potential_a", "potential_b", "marginal_a", "marginal_b"):
            v = getattr(res, k)
            if lib == "torch":
                assert isinstance(v, torch.Tensor) and str(v.dtype) == "torch." + dtype
                v = v.cpu().numpy()
            else:
                assert isinstance(v, np.ndarray) or np.isscalar(v)
                assert str(np.asarray(v).dtype) == dtype
            got[k] = v
        want = {k[5:]: c[k] for k in c if k.startswith("want_")}
        assert reference_accepts(got, want, float(c["atol"]), float(c["rtol"])) == []
        scale = max(1.0, abs(float(c["ref_value"])))
        assert abs(float(got["value"]) - float(c["ref_value"])) <= 1e-4 * scale
        assert np.abs(np.asarray(got["potential_a"], np.float64) - c["ref_potential_a"]).max() <= 1e-4 * scale


def test_correct_values_permutations(cuda):
    """Target = a shuffled copy of the source moved by a small, constant shift: at a small temperature the plan is the
    permutation matrix / N (the closed form behind generators/permutations.py), and value, plan and marginals agree with
    the float64 oracle of the same annealed loop."""
    N, D = 40, 2
    rng = np.random.default_rng(N)
    x = rng.random((N, D))
    perm = rng.permutation(N)
    y = x[perm] + 0.01
    res = ot.solve_sample(x, y, reg=1e-4, max_iter=100)
    ref = oracle_ot.solve_sample(x, y, reg=1e-4, max_iter=100)
    expected = np.zeros((N, N))
    expected[perm, np.arange(N)] = 1.0 / N
    assert np.abs(ref["plan"] - expected).max() < 1e-4 / N
    # fp32 potentials carry ~1e-7 absolute error; divided by reg = 1e-4 that is ~1e-3 relative on a plan entry
    # (the reference's own permutation test accepts atol = rtol = 5e-2, generators/permutations.py:66-69)
    assert np.abs(res.plan - expected).max() < 5e-3 / N
    assert abs(float(res.value) - ref["value"]) < 1e-4 * abs(ref["value"]) + 1e-7
    assert np.allclose(res.marginal_a, 1.0 / N, rtol=5e-3) and np.allclose(res.marginal_b, 1.0 / N, rtol=5e-3)


def test_doctest_example_of_the_reference(cuda):
    """The example in the reference's docstring (sample.py:256-279)."""
    sol = ot.solve_sample(X_a=[[0, 0], [0, 2]], X_b=[[2, 1], [2, 2]], reg=0.001, max_iter=100)
    assert np.allclose(sol.plan, [[0.5, 0.0], [0.0, 0.5]], atol=1e-3)
    assert f"{float(sol.value):.3f}" == "4.501"


def test_operators_match_the_dense_plan(cuda):
    rec = load_golden("ot_unbalanced_d2")
    x, y, a, b = (torch.from_numpy(rec[k]).float().to(cuda) for k in ("x", "y", "a", "b"))
    res = ot.solve_sample(x, y, a=a, b=b, **rec["kwargs"])
    P, dens = res.plan, res.density
    assert P.device.type == "cuda" and P.dtype == torch.float32 and P.shape == (x.shape[0], y.shape[0])
    g = torch.Generator().manual_seed(0)
    v, V = torch.randn(y.shape[0], generator=g).to(cuda), torch.randn(y.shape[0], 3, generator=g).to(cuda)
    u = torch.randn(x.shape[0], generator=g).to(cuda)
    for op, dense in ((res.plan_operator, P), (res.lazy_plan, P), (res.density_operator, dens), (res.lazy_density, dens)):
        assert op.shape == tuple(dense.shape)
        assert relerr((op @ v).cpu().numpy(), (dense @ v).cpu().numpy()) < 1e-4
        assert relerr((op @ V).cpu().numpy(), (dense @ V).cpu().numpy()) < 1e-4
        assert relerr((op.T @ u).cpu().numpy(), (dense.t() @ u).cpu().numpy()) < 1e-4
    assert relerr(res.marginal_a.cpu().numpy(), P.sum(1).cpu().numpy()) < 1e-4
    assert relerr(res.marginal_b.cpu().numpy(), P.sum(0).cpu().numpy()) < 1e-4
    with pytest.raises(ValueError, match="run your OT solver with `debias = True`"):
        res.potential_aa


def test_softmin_sample_branches(cuda):
    rng = np.random.default_rng(3)
    x, y = rng.random((50, 3)), rng.random((60, 3))
    b = rng.random(60) + 0.1
    g = rng.standard_normal(60) * 0.1
    xt, yt, lb, gt = (torch.from_numpy(v).float().to(cuda) for v in (x, y, np.log(b), g))
    C = oracle_ot.cost_matrix(x, y)
    for eps in (float("inf"), 0.3, 0.01):
        got = ot.softmin_sample(eps, lb, (xt, yt), gt).cpu().numpy()
        assert relerr(got, oracle_ot.softmin(eps, np.log(b), C, g)) < 2e-5, eps
    got = ot.softmin_sample(0, lb, (xt, yt), gt).cpu().numpy()                # hard C-transform
    assert relerr(got, oracle_ot.softmin(0, np.log(b), C, g)) < 2e-6


def test_large_problem_and_gradient(cuda):
    """Value and potentials against the float64 oracle at N = 1500, and the gradient of the value in X_a against the oracle's
    closed form (autograd sees the last soft-min only; the column cloud and the dual vector are detached)."""
    rng = np.random.default_rng(11)
    N, M = 1500, 1400
    x, y = rng.random((N, 3)), rng.random((M, 3)) * 0.7 + 0.2
    kw = dict(reg=0.02, max_iter=30)
    ref = oracle_ot.solve_sample(x, y, **kw)
    xt = torch.from_numpy(x).float().to(cuda).requires_grad_(True)
    yt = torch.from_numpy(y).float().to(cuda)
    res = ot.solve_sample(xt, yt, **kw)
    assert abs(res.value.item() - ref["value"]) <= 1e-4 * abs(ref["value"])
    assert relerr(res.potential_a.detach().cpu().numpy(), ref["potential_a"]) < 1e-4
    (gx,) = torch.autograd.grad(res.value, [xt])
    # balanced, no debias: value = <a, f_ba> + <b, g_ab>, and only f_ba's last soft-min sees x as its row cloud
    grad_ref = ref["grad_f_ba"] / N
    assert relerr(gx.cpu().numpy(), grad_ref) < 1e-4


def test_last_update_in_one_reduction(cuda, monkeypatch):
    """The last, differentiable update of `sinkhorn_loop` through hip.softmin_value_and_grad (what launches of >= 5e8 pairs
    use; forced on at N = 1500 here): same value, potentials and gradient as the forward + backward pair, and the oracle."""
    from geomloss_amd import hip

    rng = np.random.default_rng(12)
    N, M = 1500, 1300
    x, y = rng.random((N, 3)), rng.random((M, 3)) * 0.7 + 0.2
    for kw in (dict(reg=0.02, max_iter=30), dict(reg=0.05, max_iter=20, unbalanced=0.5)):
        ref = oracle_ot.solve_sample(x, y, **kw)
        res = {}
        for one_pass in (True, False):
            monkeypatch.setattr(hip, "_VALUE_GRAD_MIN_PAIRS", 0.0 if one_pass else 1e30)
            calls = []
            orig = hip.softmin_fwd_grad_raw
            monkeypatch.setattr(hip, "softmin_fwd_grad_raw", lambda *a, _o=orig, **k: (calls.append(1), _o(*a, **k))[1])
            xt = torch.from_numpy(x).float().to(cuda).requires_grad_(True)
            out = ot.solve_sample(xt, torch.from_numpy(y).float().to(cuda), **kw)
            (gx,) = torch.autograd.grad(out.value, [xt])
            res[one_pass] = (out.value.item(), out.potential_a.detach().cpu().numpy(), gx.cpu().numpy())
            monkeypatch.setattr(hip, "softmin_fwd_grad_raw", orig)
            assert bool(calls) == one_pass
        assert abs(res[True][0] - res[False][0]) <= 2e-6 * abs(res[False][0])
        assert relerr(res[True][1], res[False][1]) < 2e-6 and relerr(res[True][2], res[False][2]) < 2e-5
        assert abs(res[True][0] - ref["value"]) <= 1e-4 * abs(ref["value"])
        assert relerr(res[True][1], ref["potential_a"]) < 1e-4
