"""The reference's OWN test file for ``ot.solve_sample`` (/root/reference/tests/test_ot_solve_sample.py: Diracs drawn by
hypothesis over library / dtype / device / batch shape, checked by the reference's check_ot_result.py), run without touching a
line of it, against ``geomloss_amd.ot`` on the HIP kernels — through the import shim of tests/refshim/ (``geomloss.ot`` ->
``geomloss_amd.ot``; ``pytest_check`` stub).

Where it can run.  It needs BOTH the reference tree (read-only, present in the build container only — reference sources may not
be copied into this repository) and a GPU (``geomloss_amd.ot`` has no CPU path).  The build container has no GPU and the GPU
boxes have no reference tree, so in the driver's two test runs the GPU half is skipped; what runs everywhere the reference tree
exists is the collection half (the shim resolves every import of the reference's test package).  The same closed forms are
restated in tests/test_ot_gpu.py, which does run on the GPU box.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GEOMLOSS_REFERENCE", "/root/reference")
REF_TEST = os.path.join(REF, "tests", "test_ot_solve_sample.py")
needs_reference = pytest.mark.skipif(not os.path.exists(REF_TEST), reason="the reference tree is not on this machine")


def _pytest_on_reference(scratch, *extra):
    # nothing may be written into the reference tree: no bytecode, no pytest cache, hypothesis' example database elsewhere;
    # `-c /dev/null`: the reference's pyproject.toml asks for the pytest-cov plugin, which is not installed here
    env = dict(os.environ, GEOMLOSS_REFERENCE_SRC=os.path.join(REF, "src"), PYTHONDONTWRITEBYTECODE="1",
               HYPOTHESIS_STORAGE_DIRECTORY=os.path.join(str(scratch), "hypothesis"),
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "refshim"), ROOT, os.environ.get("PYTHONPATH", "")]))
    cmd = [sys.executable, "-m", "pytest", REF_TEST, "-q", "-c", os.devnull, "-p", "no:cacheprovider", "--rootdir", REF, *extra]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=str(scratch))


@needs_reference
def test_reference_test_file_collects_against_geomloss_amd(tmp_path):
    out = _pytest_on_reference(tmp_path, "--collect-only")
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "test_correct_values_diracs" in out.stdout


@needs_reference
@pytest.mark.gpu
def test_reference_test_file_passes_unmodified(cuda, tmp_path):
    out = _pytest_on_reference(tmp_path)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert "passed" in out.stdout
