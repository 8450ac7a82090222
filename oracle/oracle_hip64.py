"""ctypes binding of ``liboracle_hip64.so`` (``oracle_hip64.hip``): brute-force float64 reductions, one GPU thread per row.

TEST INFRASTRUCTURE ONLY: imported by ``oracle/oracle_torch64.py`` / ``tests/`` for the BASELINE sizes (1e6 x 1e6 points), never by
``geomloss_amd/``.  Inputs are anything ``torch.as_tensor`` takes; they are moved to the device as contiguous float64.  Results are
float64 CUDA tensors.  ``pattern`` = ``(lab, offsets, intervals)`` int32 CUDA tensors for block-sparse reductions (row i reduces
over ``intervals[offsets[lab[i]] : offsets[lab[i] + 1]]``), or None for all columns.
"""

import ctypes
import os

import numpy as np
import torch

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle_hip64.so")
KINDS = {"gaussian": 0, "laplacian": 1, "energy": 2}
_lib = None


def available():
    return os.path.exists(LIB_PATH) and torch.cuda.is_available()


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(LIB_PATH)
        vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        lib.o64_softmin.argtypes = [vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp]
        lib.o64_softmin_grad_x.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp]
        lib.o64_kconv.argtypes = [ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, cd, vp, vp, vp]
        for f in (lib.o64_softmin, lib.o64_softmin_grad_x, lib.o64_kconv):
            f.restype = ci
        _lib = lib
    return _lib


def _d(a, dev):
    if not isinstance(a, torch.Tensor):
        a = torch.as_tensor(np.asarray(a, np.float64))
    return a.detach().to(device=dev, dtype=torch.float64).contiguous()


def _pat(pattern):
    if pattern is None:
        return [None, None, None]
    return [t.data_ptr() for t in pattern]


def make_pattern(keep, ranges_rows, ranges_cols, dev):
    """Cluster-level keep mask (Cr, Cc) + the row ranges of the clusters in their sorted clouds -> (lab, offsets, intervals):
    row cluster k reduces over the column ranges of the clusters it keeps (one interval per kept cluster, no merging)."""
    keep = np.asarray(keep, bool)
    rr, rc = np.asarray(ranges_rows), np.asarray(ranges_cols)
    _, js = np.nonzero(keep)                                                # row-major: the intervals of a row cluster are consecutive
    offsets = np.concatenate(([0], np.cumsum(keep.sum(1)))).astype(np.int32)
    intervals = np.ascontiguousarray(rc[js]).astype(np.int32).reshape(-1, 2)
    lab = np.repeat(np.arange(len(rr), dtype=np.int32), rr[:, 1] - rr[:, 0])
    if intervals.shape[0] == 0:
        intervals = np.zeros((1, 2), np.int32)
    return tuple(torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (lab, offsets, intervals))


def softmin(eps, x, y, h, p=2, pattern=None, device="cuda:0"):
    lib = _load()
    x, y, h = _d(x, device), _d(y, device), _d(h, device).reshape(-1)
    N, D = x.shape
    M = y.shape[0]
    assert y.shape[1] == D and h.shape[0] == M and (pattern is None or pattern[0].shape[0] == N)
    out = torch.empty(N, dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        torch.cuda.synchronize()
        rc = lib.o64_softmin(x.data_ptr(), y.data_ptr(), h.data_ptr(), out.data_ptr(), N, M, D, float(eps), int(p), *_pat(pattern))
    assert rc == 0, f"o64_softmin failed ({rc})"
    return out


def softmin_grad_x(eps, x, y, h, g, p=2, pattern=None, device="cuda:0"):
    lib = _load()
    x, y, h, g = _d(x, device), _d(y, device), _d(h, device).reshape(-1), _d(g, device).reshape(-1)
    N, D = x.shape
    M = y.shape[0]
    assert y.shape[1] == D and h.shape[0] == M and g.shape[0] == N and (pattern is None or pattern[0].shape[0] == N)
    out = torch.empty((N, D), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        torch.cuda.synchronize()
        rc = lib.o64_softmin_grad_x(x.data_ptr(), y.data_ptr(), h.data_ptr(), g.data_ptr(), out.data_ptr(), N, M, D, float(eps), int(p),
                                    *_pat(pattern))
    assert rc == 0, f"o64_softmin_grad_x failed ({rc})"
    return out


def kconv(kind, x, y, v, blur=0.05, g=None, pattern=None, device="cuda:0", value=True):
    """(K v) as (N,) — and, with ``g``, d/dx sum_i g_i (K v)_i as (N, D): returns ``out``, or ``(out, grad)`` (``out`` None if not ``value``)."""
    lib = _load()
    x, y, v = _d(x, device), _d(y, device), _d(v, device).reshape(-1)
    N, D = x.shape
    M = y.shape[0]
    assert y.shape[1] == D and v.shape[0] == M
    out = torch.empty(N, dtype=torch.float64, device=x.device) if value else None
    gt = None if g is None else _d(g, device).reshape(-1)
    gout = None if g is None else torch.empty((N, D), dtype=torch.float64, device=x.device)
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with torch.cuda.device(x.device):
        torch.cuda.synchronize()
        rc = lib.o64_kconv(KINDS[kind], x.data_ptr(), y.data_ptr(), v.data_ptr(), ptr(gt), ptr(out), ptr(gout), N, M, D, float(blur),
                           *_pat(pattern))
    assert rc == 0, f"o64_kconv failed ({rc})"
    return out if g is None else (out, gout)
