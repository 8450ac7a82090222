"""Import shim used by tests/test_reference_suite.py ONLY: lets the reference's own test files — which say
``from geomloss import ot`` / ``from geomloss import _backends as bk`` — run, unmodified, against ``geomloss_amd``.

* ``geomloss.ot``        -> ``geomloss_amd.ot``  (the solver under test, on the HIP kernels)
* ``geomloss._backends`` (and the ``_typing`` module it imports) -> the reference's own array-dispatch helpers, found under
  ``$GEOMLOSS_REFERENCE_SRC`` by extending this package's search path (they are what the
  reference's checkers use to compare arrays: ``bk.allclose``, ``bk.mean`` ...; nothing of the solver lives there).  The reference
  tree only exists in the build container, so this shim — and the test that uses it — is inert anywhere else.
"""
import os
import sys

from geomloss_amd import ot  # noqa: F401  (re-exported: `from geomloss import ot`)
from geomloss_amd import SamplesLoss  # noqa: F401

_src = os.environ.get("GEOMLOSS_REFERENCE_SRC", "/root/reference/src")
_pkg = os.path.join(_src, "geomloss")
if not os.path.exists(os.path.join(_pkg, "_backends", "__init__.py")):
    raise ImportError(f"geomloss shim: the reference's _backends package was not found under {_src}")
sys.modules["geomloss.ot"] = ot          # the solver: ours.  Registered before the path below can offer the reference's.
__path__.append(_pkg)                    # `geomloss._backends`, `geomloss._typing`: the reference's array helpers, as they are
