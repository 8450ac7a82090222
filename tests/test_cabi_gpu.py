"""Callers of the C-ABI that are not ``geomloss_amd/hip.py`` (round-4 review, weak #9).

* ``tests/cabi/smoke.c``: plain C + ``hipMalloc``, compiled by gcc against ``include/glhip.h`` — dense / block-sparse soft-min,
  its gradient, the three kernel products, the no-workspace call and three error paths, checked against ``oracle_c.c`` inside
  the program.  Here: build it, run it, read its verdict.
* the ctypes stub "a geomloss maintainer would add" at the reference's own seam
  (``log_conv`` of ``_legacy/sinkhorn_samples.py:322-346,432-450``), executed VERBATIM from the document.
"""

import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, relerr

CABI = os.path.join(ROOT, "tests", "cabi")


def _build():
    """The plain-C caller needs gcc, the ROCm headers and libamdhip64 (ROCM_PATH, default /opt/rocm): a box without them skips."""
    try:
        subprocess.run(["make", "-s", "-C", CABI, "ROCM=" + os.environ.get("ROCM_PATH", "/opt/rocm")], check=True, capture_output=True)
    except (subprocess.CalledProcessError, OSError) as e:
        if not os.path.exists(os.path.join(CABI, "smoke")):
            pytest.skip(f"tests/cabi/smoke cannot be built here: {getattr(e, 'stderr', e)!r:.200}")
    return os.path.join(CABI, "smoke")


def test_c_caller_builds_and_binds_only_declared_symbols():
    """CPU: the plain-C caller compiles against include/glhip.h with -Wall -Wextra, and every glhip_* symbol it imports is
    exported by the shipped library (a prototype the library does not define would fail at link time already)."""
    exe = _build()
    und = subprocess.run(["nm", "-D", "--undefined-only", exe], check=True, capture_output=True, text=True).stdout
    want = sorted(set(re.findall(r"\bglhip_\w+", und)))
    assert {"glhip_softmin_fwd", "glhip_softmin_bwd_x", "glhip_kernel_conv_fwd", "glhip_workspace_bytes", "glhip_last_error"} <= set(want)
    lib = os.path.join(ROOT, "geomloss_amd", "libgeomloss_hip.so")
    exported = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    have = set(re.findall(r"\bglhip_\w+", exported))
    assert set(want) <= have
    dyn = subprocess.run(["readelf", "-d", lib], check=True, capture_output=True, text=True).stdout
    assert "libgeomloss_hip.so" in dyn and "SONAME" in dyn        # the bare-name dlopen of INTEGRATION.md's stub relies on it


@pytest.mark.gpu
def test_plain_c_caller_of_the_c_abi(cuda):
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    print(r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout and "FAIL" not in r.stdout
    assert r.stdout.count(" ok") >= 12


def _integration_stub():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text.split("## 2.", 1)[1]
    return re.search(r"```python\n(.*?)```", sec, re.S).group(1)


def test_integration_md_stub_is_valid_python():
    compile(_integration_stub(), "INTEGRATION.md#2", "exec")


@pytest.mark.gpu
def test_integration_md_stub_runs_verbatim(cuda):
    """The document's code block, exec'd as is.  Its ``ctypes.CDLL("libgeomloss_hip.so")`` resolves by SONAME to the library the
    package has already loaded by path (what happens inside a process that imported geomloss_amd; a standalone user puts the
    directory on LD_LIBRARY_PATH)."""
    from geomloss_amd import hip
    from geomloss_amd.cluster import from_matrix
    hip.load_library()
    ns = {}
    exec(compile(_integration_stub(), "INTEGRATION.md#2", "exec"), ns)
    g = torch.Generator().manual_seed(3)
    N, M, D = 900, 1100, 3
    x, y = torch.rand(N, D, generator=g).to(cuda), torch.rand(M, D, generator=g).to(cuda)
    h = (torch.randn(M, generator=g) - np.log(M)).to(cuda)
    for p, eps in ((2, 0.05**2), (1, 0.05)):
        log_conv = ns["log_conv_hip"](p)
        lse = log_conv(x, y, h, torch.tensor([1.0 / eps]))             # the reference passes 1/eps as a 1-element tensor (:344)
        assert lse.shape == (N, 1)
        ours = hip.softmin(eps, x, y, h, p=p)
        assert relerr((-eps * lse.view(-1)).cpu().numpy(), ours.cpu().numpy()) < 1e-6
    # block-sparse: the KeOps 6-tuple's first three entries, as `keops_lse(..., ranges=ranges_xy)` receives them (:432-450)
    ri = torch.tensor([[0, 400], [400, N]], dtype=torch.int32, device=cuda)
    rj = torch.tensor([[0, 500], [500, M]], dtype=torch.int32, device=cuda)
    keep = torch.tensor([[True, False], [True, True]], device=cuda)
    rg = from_matrix(ri, rj, keep)
    lse = ns["log_conv_hip"](2)(x, y, h, 400.0, ranges=(rg.ranges_i, rg.slices_i, rg.redranges_j))
    ours = hip.softmin(1 / 400.0, x, y, h, p=2, ranges=rg)
    assert relerr((-lse.view(-1) / 400.0).cpu().numpy(), ours.cpu().numpy()) < 1e-6
    # error path: code -> exception with the library's message
    with pytest.raises(RuntimeError, match="p must be 1 or 2"):
        ns["log_conv_hip"](3)(x, y, h, 400.0)
