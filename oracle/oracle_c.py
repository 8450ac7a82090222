"""ctypes front-end of ``liboracle_c.so`` (see ``oracle_c.c``).  TEST INFRASTRUCTURE ONLY."""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle_c.so")
_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def available():
    return os.path.exists(_PATH)


def _load():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{_PATH} is not built: run `make -C oracle`")
        _lib = ctypes.CDLL(_PATH)
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _ranges(ranges):
    """ranges = (ranges_i, slices_i, redranges_j) int arrays or None"""
    if ranges is None:
        return [None, None, None, 0], []
    keep = [np.ascontiguousarray(r, dtype=np.int32) for r in ranges]
    return [k.ctypes.data_as(_ip) for k in keep] + [int(keep[0].shape[0])], keep


def softmin(eps, x, y, h, p=2, ranges=None):
    lib = _load()
    x, px = _d(x); y, py = _d(y); h, ph = _d(h)
    N, D = x.shape; M = y.shape[0]
    out = np.full(N, np.nan)
    ra, _keep = _ranges(ranges)
    lib.oracle_softmin(px, py, ph, out.ctypes.data_as(_dp), N, M, D, ctypes.c_double(eps), int(p), *ra)
    return out


def softmin_grad_x(eps, x, y, h, g, p=2, ranges=None):
    lib = _load()
    x, px = _d(x); y, py = _d(y); h, ph = _d(h); g, pg = _d(g)
    N, D = x.shape; M = y.shape[0]
    gx = np.zeros((N, D))
    ra, _keep = _ranges(ranges)
    lib.oracle_softmin_grad_x(px, py, ph, pg, gx.ctypes.data_as(_dp), N, M, D, ctypes.c_double(eps), int(p), *ra)
    return gx


_KINDS = {"gaussian": 0, "laplacian": 1, "energy": 2}


def kconv(kind, x, y, v, blur=0.05, ranges=None):
    lib = _load()
    x, px = _d(x); y, py = _d(y); v, pv = _d(v)
    N, D = x.shape; M = y.shape[0]
    out = np.full(N, np.nan)
    ra, _keep = _ranges(ranges)
    lib.oracle_kconv(_KINDS[kind], px, py, pv, out.ctypes.data_as(_dp), N, M, D, ctypes.c_double(blur), *ra)
    return out


def kconv_grad_x(kind, x, y, v, g, blur=0.05, ranges=None):
    lib = _load()
    x, px = _d(x); y, py = _d(y); v, pv = _d(v); g, pg = _d(g)
    N, D = x.shape; M = y.shape[0]
    gx = np.zeros((N, D))
    ra, _keep = _ranges(ranges)
    lib.oracle_kconv_grad_x(_KINDS[kind], px, py, pv, pg, gx.ctypes.data_as(_dp), N, M, D, ctypes.c_double(blur), *ra)
    return gx
