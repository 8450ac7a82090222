import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


MID_SIZE = ("sinkhorn_p2_n8000", "gaussian_n8000")   # pin oracle/oracle_torch64.py; too big for the NumPy oracle


def golden_cases(mid=False):
    if mid:
        return list(MID_SIZE)
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        name = os.path.basename(f)[:-4]
        if name in ("cfg1_n2000_d2", "softmin_tensorized") or name.startswith(("images_", "volumes_", "barycenter_", "ot_", "reference_", "multiscale_", "kernel_multiscale_")):
            continue   # special cases, the grid-path vectors of make_golden_images.py (tests/test_images_*.py), the two-scale runs
        if name not in MID_SIZE:
            out.append(name)
    return out


def multiscale_cases(kernels=False):
    """Runs of the reference's own two-scale drivers (tests/golden/make_golden_multiscale.py)."""
    pre = "kernel_multiscale_" if kernels else "multiscale_"
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, pre + "*.npz")))


def ot_golden_cases():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, "ot_*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    rec = {k: z[k] for k in z.files}
    if "kwargs" in rec:
        rec["kwargs"] = eval(str(rec["kwargs"]))  # written by make_golden.py as repr(dict)
    return rec


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def reference_dirac_cases():
    """The cases drawn by the reference's own hypothesis strategy for its ``test_correct_values_diracs`` (tests/golden/
    make_golden_ot_diracs.py): dicts of NumPy arrays / scalars; absent keys mean None."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_diracs_ot.npz")
    z = np.load(path, allow_pickle=False)
    out = []
    for i in range(int(z["count"])):
        pre = f"c{i}_"
        out.append({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
    return out


def reference_accepts(got, want, atol, rtol=0.0):
    """The acceptance rule of the reference's tests/check_ot_result.py:26-83 for un-batched results, on NumPy values:
    got / want = dicts with value, plan, potential_a, potential_b, marginal_a, marginal_b.  Returns the names that fail."""
    bad = []

    def close(a, b, name, rt=rtol):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        if a.shape != b.shape or not np.allclose(a, b, atol=atol, rtol=rt, equal_nan=True):
            bad.append(name)
    close(got["value"], want["value"], "value")
    close(got["plan"], want["plan"], "plan")
    ma, mb = np.mean(got["potential_a"]), np.mean(got["potential_b"])
    wa, wb = np.mean(want["potential_a"]), np.mean(want["potential_b"])
    close(ma + mb, wa + wb, "sum(dual_potentials)", 0.0)
    close(got["potential_a"] - ma, want["potential_a"] - wa, "potential_a")
    close(got["potential_b"] - mb, want["potential_b"] - wb, "potential_b")
    for k in ("marginal_a", "marginal_b"):       # check_approx_equal skips what the expected result leaves at None
        if want.get(k) is not None:
            close(got[k], want[k], k)
    return bad
