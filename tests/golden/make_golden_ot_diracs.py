"""Generates tests/golden/reference_diracs_ot.npz: the cases that the REFERENCE's own test
``/root/reference/tests/test_ot_solve_sample.py::test_correct_values_diracs`` draws from its own hypothesis strategy
(``tests/generators/diracs.py:75-147``, derandomized), together with the closed-form answers that strategy attaches AND the outputs
of the reference's solver on them.  That test file cannot run where the GPUs are (the reference tree is not there and may not be
copied), so its drawn inputs travel instead: ``tests/test_ot_gpu.py::test_reference_suite_dirac_cases`` replays them through
``geomloss_amd.ot.solve_sample`` with the reference's own acceptance rule (``tests/check_ot_result.py``, atol = 1e-2).
Run in the build container:   python tests/golden/make_golden_ot_diracs.py
"""

import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, "/root/reference")
from hypothesis import given, settings, HealthCheck  # noqa: E402

from geomloss import ot  # noqa: E402  (the reference)
from tests import generators  # noqa: E402  (the reference's test package)
from tests.generators.common import st_method  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_diracs_ot.npz")
cases = []


def _np(v):
    if v is None:
        return None
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


@given(experiment=generators.st_diracs_sample(), method=st_method)
@settings(deadline=None, max_examples=60, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
def draw(experiment, method):
    ex = experiment
    res = ot.solve_sample(ex.X_a, ex.X_b, a=ex.a, b=ex.b, cost=ex.cost, reg=ex.reg, unbalanced=ex.unbalanced, max_iter=ex.max_iter,
                          method=method)
    library = "torch" if hasattr(ex.X_a, "detach") else "numpy"
    cases.append(dict(
        X_a=_np(ex.X_a), X_b=_np(ex.X_b), a=_np(ex.a), b=_np(ex.b), reg=float(ex.reg), max_iter=int(ex.max_iter), atol=float(ex.atol),
        rtol=float(ex.rtol), library=library, dtype=str(_np(ex.X_a).dtype), method=method,
        want_value=_np(ex.result.value), want_plan=_np(ex.result.plan), want_potential_a=_np(ex.result.potential_a),
        want_potential_b=_np(ex.result.potential_b), want_marginal_a=_np(ex.result.marginal_a), want_marginal_b=_np(ex.result.marginal_b),
        ref_value=_np(res.value), ref_plan=_np(res.plan), ref_potential_a=_np(res.potential_a), ref_potential_b=_np(res.potential_b)))


def main():
    draw()
    rec = {"count": np.int64(len(cases))}
    for i, c in enumerate(cases):
        for k, v in c.items():
            if v is not None:
                rec[f"c{i}_{k}"] = np.asarray(v)
    np.savez_compressed(OUT, **rec)
    dims = sorted({c["X_a"].shape[-1] for c in cases})
    print(f"{len(cases)} cases, dimensions {dims}, libraries {sorted({c['library'] for c in cases})}, "
          f"dtypes {sorted({c['dtype'] for c in cases})}, weights given in {sum(c['a'] is not None for c in cases)} of them")


if __name__ == "__main__":
    main()
