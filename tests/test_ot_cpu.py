"""``geomloss_amd.ot`` without a GPU: the NumPy oracle of ``ot.solve_sample`` pinned to outputs of the reference itself
(tests/golden/ot_*.npz, written by tests/golden/make_golden_ot.py), the annealing schedule, the argument checks (same
exceptions and messages as the reference) and the refusal to run without a GPU."""

import numpy as np
import pytest
import torch

from conftest import load_golden, ot_golden_cases, reference_accepts, reference_dirac_cases, relerr
from geomloss_amd import ot
from oracle import oracle_ot


def _inputs(rec):
    kw = dict(rec["kwargs"])
    if "blur" in kw:
        kw["reg"] = 2 * kw.pop("blur") ** 2
    if "reach" in kw:
        kw["unbalanced"] = 2 * kw.pop("reach") ** 2
    return rec["x"], rec["y"], rec.get("a"), rec.get("b"), kw


@pytest.mark.parametrize("name", ot_golden_cases())
def test_ot_oracle_matches_reference(name):
    rec = load_golden(name)
    x, y, a, b, kw = _inputs(rec)
    out = oracle_ot.solve_sample(x, y, a, b, **kw)
    assert abs(out["value"] - float(rec["value"])) <= 1e-10 * abs(float(rec["value"]))
    for k in ("potential_a", "potential_b", "marginal_a", "marginal_b", "potential_aa", "potential_bb", "plan"):
        if k in rec:
            assert relerr(out[k], rec[k]) < 1e-9, k
    if "plan_rows" in rec:
        assert relerr(out["plan"][:5], rec["plan_rows"]) < 1e-9


def test_reference_dirac_cases_pin_the_oracle():
    """The 60 cases the reference's own ``test_correct_values_diracs`` draws (its hypothesis strategy, derandomized:
    tests/golden/make_golden_ot_diracs.py) with the outputs of the reference's solver on them: (i) the reference passes its own
    acceptance rule on its own cases — the stored closed forms and the stored outputs are consistent; (ii) the float64 oracle
    reproduces the reference's outputs."""
    cases = reference_dirac_cases()
    assert len(cases) == 60 and {c["X_a"].shape[-1] for c in cases} == {1, 2, 3, 4, 5}
    for c in cases:
        want = {k[5:]: c[k] for k in c if k.startswith("want_")}
        ref = {k[4:]: c[k] for k in c if k.startswith("ref_")}
        assert reference_accepts(ref, want, float(c["atol"]), float(c["rtol"])) == []
        out = oracle_ot.solve_sample(c["X_a"].astype(np.float64), c["X_b"].astype(np.float64), a=c.get("a"), b=c.get("b"),
                                     reg=float(c["reg"]), max_iter=int(c["max_iter"]))
        tol = 1e-9 if str(c["dtype"]) == "float64" else 2e-5       # half of the cases ran in float32 in the reference
        scale = max(1.0, abs(float(c["ref_value"])))
        assert abs(out["value"] - float(c["ref_value"])) <= tol * scale
        assert np.abs(out["potential_a"] - c["ref_potential_a"]).max() <= tol * scale
        # the reference's float32 plan exp((f + g - C) / reg) carries the cancellation its own strategy warns about (diracs.py:80-83)
        assert np.abs(out["plan"] - c["ref_plan"]).max() <= (1e-6 if str(c["dtype"]) == "float64" else float(c["atol"]))


def test_annealing_parameters():
    d = ot.annealing_parameters(maxmin_cost=4.0, eps=0.01, n_iter=5)
    assert np.allclose(d.eps_list, np.geomspace(4.0, 0.01, 5)) and d.scale_list == [0] * 5 and d.rho_list == [None] * 5
    assert ot.annealing_parameters(maxmin_cost=4.0, eps=0.01, n_iter=1).eps_list == [0.01]
    assert ot.annealing_parameters(maxmin_cost=1e-3, eps=0.5, n_iter=3).eps_list == [0.5, 0.5, 0.5]      # diameter < blur
    d = ot.annealing_parameters(maxmin_cost=1.0, eps=0.01, scaling=0.1)                                    # annealing.py:150-153
    assert np.allclose(d.eps_list, [1.0, 0.1, 0.01, 0.01])
    assert ot.annealing_parameters(maxmin_cost=1.0, eps=0.2, n_iter=3, scaling=1).eps_list == [0.2] * 3
    d = ot.annealing_parameters(maxmin_cost=1.0, eps=0.05, rho=2.0, n_iter=7, eps_scales=[1.0, 0.5, 0.1])
    assert d.rho_list == [2.0] * 7 and d.scale_list[-1] == 2 and sorted(d.scale_list) == d.scale_list
    for bad in (dict(n_iter=0), dict(scaling=1.5), dict(), dict(scaling=1)):
        with pytest.raises(ValueError):
            ot.annealing_parameters(maxmin_cost=1.0, eps=0.1, **bad)


@pytest.mark.parametrize("kwargs,exc,msg", [
    (dict(reg=-1.0, max_iter=3), ValueError, "Parameter 'reg' should be >= 0"),
    (dict(reg=0, max_iter=3), NotImplementedError, "we require that reg > 0"),
    (dict(reg=0.1, unbalanced=-2.0, max_iter=3), ValueError, "Parameter 'unbalanced' should be None"),
    (dict(reg=0.1, unbalanced=1.0, unbalanced_type="TV", max_iter=3), NotImplementedError, "'KL' penalty"),
    (dict(reg=0.1, method="lbfgs", max_iter=3), NotImplementedError, "single method"),
    (dict(reg=0.1), ValueError, "'max_iter' parameter should be a positive integer"),
    (dict(reg=0.1, max_iter=3, tol=1e-3), NotImplementedError, "stopping criteria"),
    (dict(reg=0.1, blur=0.1, max_iter=3), ValueError, "'reg' and 'blur' are redundant"),
    (dict(reg=0.1, unbalanced=1.0, reach=0.1, max_iter=3), ValueError, "'unbalanced' and 'reach' are redundant"),
    (dict(reg=0.1, max_iter=3, cost="euclidean"), NotImplementedError, None),
])
def test_solver_parameter_checks(kwargs, exc, msg):
    with pytest.raises(exc, match=msg):
        ot.solve_sample([[0.0, 0.0], [0.0, 2.0]], [[2.0, 1.0], [2.0, 2.0]], **kwargs)


def test_input_checks():
    x, y = np.zeros((3, 2)), np.ones((4, 2))
    kw = dict(reg=0.1, max_iter=2)
    with pytest.raises(ValueError, match=r"Expected X_a to be a \(N, D\) array"):
        ot.solve_sample(np.zeros(3), y, **kw)
    with pytest.raises(ValueError, match=r"Expected X_b to be a \(M, D\) array"):
        ot.solve_sample(x, np.zeros((2, 4, 2)), **kw)
    with pytest.raises(ValueError, match="same number of coordinates per sample"):
        ot.solve_sample(x, np.ones((4, 3)), **kw)
    with pytest.raises(ValueError, match="The marginal 'a' should be of shape"):
        ot.solve_sample(x, y, a=np.ones(5) / 5, **kw)
    with pytest.raises(ValueError, match="The marginal 'b' contains negative values"):
        ot.solve_sample(x, y, b=np.array([0.5, 0.5, 0.5, -0.5]), **kw)
    with pytest.raises(ValueError, match="do not sum up to the same value"):
        ot.solve_sample(x, y, a=np.ones(3), b=np.ones(4), **kw)
    with pytest.raises(ValueError, match="same tensor library"):
        ot.solve_sample(x, torch.ones(4, 2, dtype=torch.float64), **kw)
    with pytest.raises(ValueError, match="same numerical dtype"):
        ot.solve_sample(x, y.astype(np.float32), **kw)


def test_no_gpu_means_no_result():
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    with pytest.raises(RuntimeError, match="no GPU"):
        ot.solve_sample(np.zeros((3, 2)), np.ones((4, 2)), reg=0.1, max_iter=2)


def test_out_of_scope_entry_points_say_so():
    for name in ("solve", "solve_batch", "barycenter", "solve_grid", "barycenter_grid", "barycenter_sample"):
        with pytest.raises(NotImplementedError):
            getattr(ot, name)()
    with pytest.raises(NotImplementedError):
        ot.solve_sample_batch()


def test_linear_operator_algebra():
    A = torch.randn(5, 7)
    op = ot.LinearOperator(matmat=lambda s: A @ s, rmatmat=lambda s: A.t() @ s, input_shape=(7,), output_shape=(5,))
    v, V = torch.randn(7), torch.randn(7, 3)
    assert op.shape == (5, 7) and op.T.shape == (7, 5)
    assert torch.allclose(op @ v, A @ v) and torch.allclose(op @ V, A @ V) and torch.allclose(op.T @ torch.ones(5), A.t() @ torch.ones(5))
    r, c = torch.rand(5), torch.rand(7)
    sc = op.rescale(input_scaling=c, output_scaling=r)
    assert torch.allclose(sc @ v, r * (A @ (c * v)), atol=1e-6) and torch.allclose(sc.T @ torch.ones(5), c * (A.t() @ r), atol=1e-6)
    with pytest.raises(ValueError):
        op @ torch.randn(6)
