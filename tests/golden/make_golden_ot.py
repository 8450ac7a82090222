"""Generates tests/golden/ot_*.npz by running the REFERENCE ``geomloss.ot.solve_sample`` (jeanfeydy/geomloss 0.3.1 at
/root/reference, dense-cost branch: pykeops is not installed) on seeded inputs, in float64 NumPy.  Run in the build container:

    python tests/golden/make_golden_ot.py

Only this script reads /root/reference; tests read the committed .npz files (tests/test_ot_cpu.py pins the oracle,
tests/test_ot_gpu.py checks the HIP solver)."""

import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/src")
from geomloss import ot  # noqa: E402  (the reference)

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (seed, N, M, D, random weights, solver kwargs)
    "ot_balanced_d2": (1, 60, 70, 2, True, dict(reg=0.05, max_iter=30)),
    "ot_balanced_d3_uniform": (2, 90, 80, 3, False, dict(reg=0.02, max_iter=40)),
    "ot_debias_d3": (3, 70, 75, 3, True, dict(reg=0.05, max_iter=30, debias=True)),
    "ot_unbalanced_d2": (4, 65, 55, 2, True, dict(reg=0.05, unbalanced=0.5, max_iter=30)),
    "ot_unbalanced_debias_d1": (5, 50, 60, 1, True, dict(reg=0.02, unbalanced=1.0, max_iter=25, debias=True)),
    "ot_few_iterations_d5": (6, 40, 45, 5, True, dict(reg=0.3, max_iter=3)),
    "ot_one_iteration_d3": (7, 30, 35, 3, True, dict(reg=0.5, max_iter=1)),
    "ot_blur_reach_d3": (8, 80, 85, 3, True, dict(blur=0.1, reach=0.7, max_iter=30)),
    "ot_mid_d3": (9, 700, 650, 3, True, dict(reg=0.01, max_iter=40)),
}


def main():
    for name, (seed, N, M, D, wts, kw) in CASES.items():
        rng = np.random.default_rng(seed)
        x, y = rng.random((N, D)), rng.random((M, D)) * 0.8 + 0.15
        a = b = None
        if wts:
            a, b = rng.random(N) + 0.1, rng.random(M) + 0.1
            if "unbalanced" in kw or "reach" in kw:
                a, b = a / a.sum(), 1.3 * b / b.sum()           # different total masses
            else:
                a, b = a / a.sum(), b / b.sum()
        res = ot.solve_sample(x, y, a=a, b=b, **kw)
        rec = dict(x=x, y=y, kwargs=repr(kw), value=res.value, potential_a=res.potential_a, potential_b=res.potential_b,
                   marginal_a=res.marginal_a, marginal_b=res.marginal_b)
        if a is not None:
            rec.update(a=a, b=b)
        if kw.get("debias"):
            rec.update(potential_aa=res.potential_aa, potential_bb=res.potential_bb)
        if N * M <= 10000:
            rec["plan"] = res.plan
        else:   # a few rows of the plan are enough for the big case
            rec["plan_rows"] = res.plan[:5]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, "value", float(res.value))


if __name__ == "__main__":
    main()
