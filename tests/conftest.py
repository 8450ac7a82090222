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
        if name in ("cfg1_n2000_d2", "softmin_tensorized") or name.startswith(("images_", "volumes_", "barycenter_", "ot_")):
            continue   # special cases, and the grid-path vectors of make_golden_images.py (tests/test_images_*.py)
        if name not in MID_SIZE:
            out.append(name)
    return out


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
