"""ctypes binding of ``libgeomloss_hip.so`` (C-ABI: ``include/glhip.h``) and the autograd
functions built on it.

This module is the only place where Python touches the HIP kernels.  It plays the role that
``pykeops.torch`` plays for the reference (``generic_logsumexp`` at
``_legacy/sinkhorn_samples.py:322-334,432-442``; ``LazyTensor @ v`` at
``_legacy/kernel_samples.py:128-137``).  There is no fallback: if the shared library is missing or
a tensor is not on a GPU, the call raises.
"""

import ctypes
import os
import warnings

import torch
from torch.autograd.function import once_differentiable

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgeomloss_hip.so")

GAUSSIAN, LAPLACIAN, ENERGY = 0, 1, 2
KERNEL_KINDS = {"gaussian": GAUSSIAN, "laplacian": LAPLACIAN, "energy": ENERGY}
F32, BF16 = 0, 1
FLAG_DIRECT, FLAG_NO_MFMA, FLAG_NO_SPLIT, FLAG_F32_MFMA, FLAG_XDL16, FLAG_PREPACK, FLAG_MFMA_DIST, FLAG_SMALL_ROW_BLOCKS = 1, 2, 4, 8, 16, 32, 64, 128
FLAG_F16X2 = 256                # exponents from two f16 pieces per coordinate; the caller vouches for the range (glhip.h)
FLAG_NO_SORT = 512              # big dense distance reductions: do not voxel-sort the clouds inside the library (glhip.h)
FLAG_GRAD_FAMILY = FLAG_XDL16   # kernel products rounded like the product-and-gradient kernel of the same kind (glhip.h)
XD_MAX_DIM = 16                 # p = 2 soft-min forward / half-step and gaussian product run on the matrix cores up to this dimension

# every symbol include/glhip.h declares, with its ctypes signature
_c_int, _c_float, _vp, _c_size = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
_c_long = ctypes.c_long
_RANGES = [_vp, _vp, _vp, _c_int]
_TAIL = [_vp, _c_size, _c_int, _vp]  # workspace, workspace_bytes, flags, stream
SIGNATURES = {
    "glhip_version": (_c_int, []),
    "glhip_last_error": (ctypes.c_char_p, []),
    "glhip_workspace_bytes": (_c_size, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "glhip_softmin_fwd": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _c_int]
                          + _RANGES + _TAIL),
    "glhip_sinkhorn_step": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float,
                                     _c_int, _c_int] + _RANGES + _TAIL),
    "glhip_sinkhorn_iter4": (_c_int, [_vp] * 12 + [_c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_int, _c_int, _c_int]
                             + _TAIL),
    "glhip_sinkhorn_anneal": (_c_int, [_vp] * 6 + [_c_int] * 4 + [_vp, _vp, _c_int, _c_int, _c_int, _vp, _c_size, _c_int, _c_float, _vp]),
    "glhip_sinkhorn_extrapolate4": (_c_int, [_vp] * 14 + [_c_int] * 6 + [_c_float, _c_float, _c_int, _c_int] + _TAIL),
    "glhip_softmin_bwd_x": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int,
                                     _c_int] + _RANGES + _TAIL),
    "glhip_kernel_conv_fwd": (_c_int, [_c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int]
                              + _RANGES + _TAIL),
    "glhip_kernel_conv_bwd_x": (_c_int, [_c_int, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float,
                                         _c_int] + _RANGES + _TAIL),
    "glhip_softmin_fwd_grad": (_c_int, [_vp, _vp, _vp, _vp, _c_float, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int,
                                        _c_int] + _RANGES + _TAIL),
    "glhip_kernel_conv_fwd_grad": (_c_int, [_c_int, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float,
                                            _c_int] + _RANGES + _TAIL),
    "glhip_softmin_dense_fwd": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_float, _vp]),
    "glhip_bounding_box": (_c_int, [_vp, _c_long, _vp, _c_long, _c_int, _c_int, _vp, _vp]),
    "glhip_log_weights": (_c_int, [_vp, _vp, _vp, _c_int, _vp]),
    "glhip_sinkhorn_cost": (_c_int, [_vp] * 7 + [_c_int] * 5 + [_vp]),
    "glhip_lse_lines_fwd": (_c_int, [_vp, _vp, ctypes.c_long, _c_int, _c_float, _c_int, _vp]),
    "glhip_lse_lines_bwd": (_c_int, [_vp, _vp, _vp, _vp, ctypes.c_long, _c_int, _c_float, _c_int, _vp]),
    "glhip_cmin_fwd": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int] + _RANGES + _TAIL),
    "glhip_max_lines_fwd": (_c_int, [_vp, _vp, ctypes.c_long, _c_int, _c_float, _c_int, _vp]),
    "glhip_cluster_workspace_bytes": (_c_size, [_c_int, _c_int]),
    "glhip_grid_cluster": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_float, _c_float] + [_vp] * 7 + [_vp, _c_size, _vp]),
    "glhip_block_ranges": (_c_int, [_c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float] + [_vp] * 6
                           + [ctypes.c_longlong, _vp, _vp]),
    "glhip_block_ranges_count": (_c_int, [_c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float] + [_vp] * 5 + [_vp]),
    "glhip_block_ranges_kept_pairs": (_c_int, [_c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float] + [_vp] * 4),
}
_c_double = ctypes.c_double
SIGNATURES.update({
    # float64 reductions (glhip_api_f64.hip): every array is double; no workspace, no flags
    "glhip_softmin_fwd_f64": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_double, _c_int] + _RANGES + [_vp]),
    "glhip_sinkhorn_step_f64": (_c_int, [_vp] * 6 + [_c_int, _c_int, _c_int, _c_int, _c_double, _c_double, _c_int] + _RANGES + [_vp]),
    "glhip_softmin_bwd_x_f64": (_c_int, [_vp] * 6 + [_c_int, _c_int, _c_int, _c_int, _c_double, _c_int] + _RANGES + [_vp]),
    "glhip_kernel_conv_fwd_f64": (_c_int, [_c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_double] + _RANGES + [_vp]),
    "glhip_kernel_conv_bwd_x_f64": (_c_int, [_c_int] + [_vp] * 5 + [_c_int, _c_int, _c_int, _c_int, _c_double] + _RANGES + [_vp]),
})
KEEP_DUAL_SLACK, KEEP_WITHIN = 0, 1

_lib = None


def load_library(path=None):
    """Loads (once) and returns the shared library; raises ``RuntimeError`` if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("GEOMLOSS_HIP_LIB") or LIB_PATH      # GEOMLOSS_HIP_LIB: another build of the same C-ABI (A/B runs)
    if not os.path.exists(path):
        raise RuntimeError(
            f"geomloss_amd: the HIP extension {path} is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C geomloss_amd/csrc`). "
            "The 'online' and 'multiscale' backends have no CPU or PyTorch fallback."
        )
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = restype, argtypes
    _lib = lib
    return lib


def library_available():
    return os.path.exists(LIB_PATH)


def _check(rc, lib):
    if rc != 0:
        msg = lib.glhip_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        if rc == -2:
            raise NotImplementedError(msg)
        raise RuntimeError(msg)


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _points(t, name, allow_f64=False):
    """Point clouds go to the kernels as contiguous fp32 or bf16 — or, from the four entry points that have double-precision kernels
    (``allow_f64``: soft-min, its gradient, kernel product, its gradient), as fp64; every other float type is widened / narrowed
    to fp32.  float64 clouds keep their dtype there — the reference's matrix-free backends do (``_legacy/sinkhorn_samples.py:229-290``)
    — and run on ``glhip_*_f64`` (any D; no matrix cores, ~20x slower than fp32: cast to fp32 for speed).  The fp32-only entry points
    (one-launch iteration, one-pass value + gradient, hard C-transform) never see an fp64 buffer: a launch with dtype code F32 on
    doubles would read garbage silently (round-4 advice)."""
    if not t.is_cuda:
        raise RuntimeError(
            f"geomloss_amd: '{name}' lives on {t.device}; the HIP backends ('online', 'multiscale') need GPU tensors. "
            "Use backend='tensorized' for CPU tensors."
        )
    if t.dtype not in ((torch.float32, torch.bfloat16, torch.float64) if allow_f64 else (torch.float32, torch.bfloat16)):
        t = t.float()
    return t.contiguous()


def is_f64(t):
    return t.dtype == torch.float64


def _dtype_code(t):
    return BF16 if t.dtype == torch.bfloat16 else F32


def _f32(t):
    return t.detach().float().contiguous()


def _acc_dtype(points):
    """dtype of dual vectors, weights, gradients and outputs next to clouds of this dtype: fp64 with fp64 clouds, fp32 otherwise."""
    return torch.float64 if points.dtype == torch.float64 else torch.float32


def _vec(t, points):
    """A per-point vector (dual values, weights, gradients) in the accumulation dtype of the clouds it goes with."""
    return t.detach().to(_acc_dtype(points)).contiguous()


class BlockRanges:
    """Block-sparse reduction pattern in the KeOps convention (see ``include/glhip.h``), both orientations.

    Stands in for the 6-tuple returned by ``pykeops.torch.cluster.from_matrix``
    (``_legacy/sinkhorn_samples.py:515``); ``.t()`` is ``swap_axes`` (``:529``).
    """

    def __init__(self, ranges_i, slices_i, redranges_j, ranges_j, slices_j, redranges_i, small_i=False, small_j=False):
        self.ranges_i, self.slices_i, self.redranges_j = ranges_i, slices_i, redranges_j
        self.ranges_j, self.slices_j, self.redranges_i = ranges_j, slices_j, redranges_i
        # launch hints (GLHIP_FLAG_SMALL_ROW_BLOCKS): the pairs of this orientation / of the transposed one sit in row blocks of
        # up to 64 points.  Set by whoever knows the block sizes (sinkhorn_samples.kernel_truncation); results do not depend on them.
        self.small_i, self.small_j = bool(small_i), bool(small_j)

    def t(self):
        return BlockRanges(self.ranges_j, self.slices_j, self.redranges_i,
                           self.ranges_i, self.slices_i, self.redranges_j, self.small_j, self.small_i)

    def launch_flags(self):
        return FLAG_SMALL_ROW_BLOCKS if self.small_i else 0

    def c_args(self):
        n = int(self.ranges_i.shape[0])
        return [ctypes.c_void_p(self.ranges_i.data_ptr()), ctypes.c_void_p(self.slices_i.data_ptr()),
                ctypes.c_void_p(self.redranges_j.data_ptr()), n]


_NO_RANGES = [None, None, None, 0]


def _range_flags(flags, ranges):
    return int(flags) | (0 if ranges is None else ranges.launch_flags())


def _range_args(ranges, B):
    if ranges is None:
        return _NO_RANGES
    if B != 1:
        raise NotImplementedError("Block-sparse reductions are only implemented for a single (un-batched) problem.")
    return ranges.c_args()


def _workspace(lib, x, B, N, M, D, ranges):
    """Scratch buffer for column splits (torch's caching allocator makes this a free-list lookup)."""
    n_ranges = 0 if ranges is None else int(ranges.ranges_i.shape[0])
    nbytes = int(lib.glhip_workspace_bytes(B, N, M, D, n_ranges))
    if nbytes == 0:
        return None, [None, 0]
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    return ws, [ctypes.c_void_p(ws.data_ptr()), nbytes]


def _as_batched(x, y, s):
    """(N,D),(M,D),(M,) -> (1,N,D),(1,M,D),(1,M); batched inputs pass through.  The per-column vector follows the clouds' accumulation
    dtype (fp64 next to fp64 clouds)."""
    if s.dtype != _acc_dtype(x):
        s = s.to(_acc_dtype(x))
    if x.dim() == 2:
        return x.unsqueeze(0), y.unsqueeze(0), s.reshape(1, -1), False
    return x, y, s.reshape(x.shape[0], -1), True


# ----------------------------------------------------------------------------------------------
#  raw launches
# ----------------------------------------------------------------------------------------------

def softmin_fwd_raw(x, y, h, eps, p=2, ranges=None, flags=0):
    """x (B,N,D), y (B,M,D) fp32|bf16 contiguous CUDA; h (B,M) fp32 -> (B,N) fp32."""
    lib = load_library()
    B, N, D = x.shape
    M = y.shape[1]
    if is_f64(x):
        out = torch.empty((B, N), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.glhip_softmin_fwd_f64(x.data_ptr(), y.data_ptr(), h.data_ptr(), out.data_ptr(), B, N, M, D, float(eps), int(p),
                                           *_range_args(ranges, B), _stream(x))
        _check(rc, lib)
        return out
    out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws, ws_args = _workspace(lib, x, B, N, M, D, ranges)
        rc = lib.glhip_softmin_fwd(x.data_ptr(), y.data_ptr(), h.data_ptr(), out.data_ptr(), B, N, M, D,
                                   float(eps), int(p), _dtype_code(x), *_range_args(ranges, B), *ws_args,
                                   _range_flags(flags, ranges), _stream(x))
    _check(rc, lib)
    return out


def sinkhorn_step_raw(x, y, logw, pot, prev, eps, damping, p=2, ranges=None, flags=0):
    """Fused half-step: (prev + damping * softmin(eps, C(x,y), logw + pot/eps)) / 2, or damping * softmin(...) if prev is None."""
    lib = load_library()
    B, N, D = x.shape
    M = y.shape[1]
    if is_f64(x):      # every array double (glhip_sinkhorn_step_f64): no workspace, no flags
        out = torch.empty((B, N), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.glhip_sinkhorn_step_f64(x.data_ptr(), y.data_ptr(), logw.data_ptr(), None if pot is None else pot.data_ptr(),
                                             None if prev is None else prev.data_ptr(), out.data_ptr(), B, N, M, D, float(eps),
                                             float(damping), int(p), *_range_args(ranges, B), _stream(x))
        _check(rc, lib)
        return out
    out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws, ws_args = _workspace(lib, x, B, N, M, D, ranges)
        rc = lib.glhip_sinkhorn_step(x.data_ptr(), y.data_ptr(), logw.data_ptr(),
                                     None if pot is None else pot.data_ptr(), None if prev is None else prev.data_ptr(),
                                     out.data_ptr(), B, N, M, D, float(eps), float(damping), int(p), _dtype_code(x),
                                     *_range_args(ranges, B), *ws_args, _range_flags(flags, ranges), _stream(x))
    _check(rc, lib)
    return out


def sinkhorn_iter4_raw(x, y, a_log, b_log, pots, eps, damping, debias=True, flags=0, p=2):
    """The 4 (or 2) simultaneous updates of one Sinkhorn iteration in one launch.

    x (B,N,D), y (B,M,D); a_log (B,N), b_log (B,M); ``pots`` = None (initialisation) or the old potentials
    ``(f_ba, g_ab, f_aa, g_bb)`` / ``(f_ba, g_ab)`` as (B,N) / (B,M) fp32.  Returns the new ones, same arity."""
    lib = load_library()
    B, N, D = x.shape
    M = y.shape[1]
    first = pots is None
    outs = [torch.empty((B, n), dtype=torch.float32, device=x.device) for n in ((N, M, N, M) if debias else (N, M))]
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    old = [None] * 4 if first else list(pots) + [None] * (4 - len(pots))
    new = outs + [None] * (4 - len(outs))
    with torch.cuda.device(x.device):
        L = max(N, M)
        nbytes = 4 * int(lib.glhip_workspace_bytes(B, L, L, D, 0))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
        rc = lib.glhip_sinkhorn_iter4(x.data_ptr(), y.data_ptr(), a_log.data_ptr(), b_log.data_ptr(),
                                      *[ptr(t) for t in old], *[ptr(t) for t in new], B, N, M, D, float(eps), float(damping),
                                      int(p), _dtype_code(x), int(first), ptr(ws), nbytes, int(flags), _stream(x))
    _check(rc, lib)
    return tuple(outs)


def sinkhorn_extrapolate4_raw(x, y, xc, yc, a_log_c, b_log_c, pots, eps, damping, flags=0, p=2):
    """The coarse-to-fine jump in one launch (``glhip_sinkhorn_extrapolate4``): fine clouds x (B,N,D), y (B,M,D), coarse clouds
    xc (B,Nc,D), yc (B,Mc,D) with log-weights a_log_c (B,Nc), b_log_c (B,Mc) and potentials ``pots`` = (f_ba, g_ab, f_aa, g_bb) /
    (f_ba, g_ab) on the coarse clouds.  Returns the potentials on the fine clouds, same arity."""
    lib = load_library()
    B, N, D = x.shape
    M, Nc, Mc = y.shape[1], xc.shape[1], yc.shape[1]
    debias = len(pots) == 4
    outs = [torch.empty((B, n), dtype=torch.float32, device=x.device) for n in ((N, M, N, M) if debias else (N, M))]
    old = [t.data_ptr() for t in pots] + [None] * (4 - len(pots))
    new = [t.data_ptr() for t in outs] + [None] * (4 - len(outs))
    with torch.cuda.device(x.device):
        nbytes = 4 * int(lib.glhip_workspace_bytes(B, max(N, M), max(Nc, Mc), D, 0))     # rows: fine clouds, columns: coarse ones
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
        rc = lib.glhip_sinkhorn_extrapolate4(x.data_ptr(), y.data_ptr(), xc.data_ptr(), yc.data_ptr(), a_log_c.data_ptr(), b_log_c.data_ptr(),
                                             *old, *new, B, N, M, Nc, Mc, D, float(eps), float(damping), int(p), _dtype_code(x),
                                             None if ws is None else ws.data_ptr(), nbytes, int(flags), _stream(x))
    _check(rc, lib)
    return tuple(outs)


def softmin_bwd_x_raw(x, y, h, out, grad_out, eps, p=2, ranges=None, flags=0):
    lib = load_library()
    B, N, D = x.shape
    M = y.shape[1]
    if is_f64(x):
        gx = torch.empty((B, N, D), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.glhip_softmin_bwd_x_f64(x.data_ptr(), y.data_ptr(), h.data_ptr(), out.data_ptr(), grad_out.data_ptr(), gx.data_ptr(),
                                             B, N, M, D, float(eps), int(p), *_range_args(ranges, B), _stream(x))
        _check(rc, lib)
        return gx
    gx = torch.empty((B, N, D), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws, ws_args = _workspace(lib, x, B, N, M, D, ranges)
        rc = lib.glhip_softmin_bwd_x(x.data_ptr(), y.data_ptr(), h.data_ptr(), out.data_ptr(), grad_out.data_ptr(),
                                     gx.data_ptr(), B, N, M, D, float(eps), int(p), _dtype_code(x),
                                     *_range_args(ranges, B), *ws_args, int(flags), _stream(x))
    _check(rc, lib)
    return gx


def softmin_fwd_grad_raw(x, y, h, guess, margin, eps, ranges=None, flags=0):
    """Soft-min and d out_i / d x_i in one reduction, given a guess within ``margin`` of the answer (``glhip_softmin_fwd_grad``;
    p = 2, D <= 16) -> (B,N), (B,N,D)."""
    lib = load_library()
    B, N, D = x.shape
    M = y.shape[1]
    out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    gu = torch.empty((B, N, D), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws, ws_args = _workspace(lib, x, B, N, M, D, ranges)
        rc = lib.glhip_softmin_fwd_grad(x.data_ptr(), y.data_ptr(), h.data_ptr(), guess.data_ptr(), float(margin), out.data_ptr(),
                                        gu.data_ptr(), B, N, M, D, float(eps), 2, _dtype_code(x), *_range_args(ranges, B), *ws_args,
                                        int(flags), _stream(x))
    _check(rc, lib)
    return out, gu


def kernel_conv_fwd_raw(kind, x, y, v, blur, ranges=None, flags=0):
    lib = load_library()
    B, N, D = x.shape
    M = y.shape[1]
    if is_f64(x):
        out = torch.empty((B, N), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.glhip_kernel_conv_fwd_f64(int(kind), x.data_ptr(), y.data_ptr(), v.data_ptr(), out.data_ptr(), B, N, M, D, float(blur),
                                               *_range_args(ranges, B), _stream(x))
        _check(rc, lib)
        return out
    out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws, ws_args = _workspace(lib, x, B, N, M, D, ranges)
        rc = lib.glhip_kernel_conv_fwd(int(kind), x.data_ptr(), y.data_ptr(), v.data_ptr(), out.data_ptr(), B, N, M, D,
                                       float(blur), _dtype_code(x), *_range_args(ranges, B), *ws_args, int(flags),
                                       _stream(x))
    _check(rc, lib)
    return out


def kernel_conv_bwd_x_raw(kind, x, y, v, g, blur, ranges=None, flags=0):
    lib = load_library()
    B, N, D = x.shape
    M = y.shape[1]
    if is_f64(x):
        gx = torch.empty((B, N, D), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.glhip_kernel_conv_bwd_x_f64(int(kind), x.data_ptr(), y.data_ptr(), v.data_ptr(), g.data_ptr(), gx.data_ptr(), B, N, M, D,
                                                 float(blur), *_range_args(ranges, B), _stream(x))
        _check(rc, lib)
        return gx
    gx = torch.empty((B, N, D), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws, ws_args = _workspace(lib, x, B, N, M, D, ranges)
        rc = lib.glhip_kernel_conv_bwd_x(int(kind), x.data_ptr(), y.data_ptr(), v.data_ptr(), g.data_ptr(),
                                         gx.data_ptr(), B, N, M, D, float(blur), _dtype_code(x),
                                         *_range_args(ranges, B), *ws_args, int(flags), _stream(x))
    _check(rc, lib)
    return gx


def kernel_conv_fwd_grad_raw(kind, x, y, v, blur, ranges=None, flags=0):
    """out = K v and d out_i / d x_i in one pass (``glhip_kernel_conv_fwd_grad``; gaussian, D <= 3) -> (B,N), (B,N,D)."""
    lib = load_library()
    B, N, D = x.shape
    M = y.shape[1]
    out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    gu = torch.empty((B, N, D), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws, ws_args = _workspace(lib, x, B, N, M, D, ranges)
        rc = lib.glhip_kernel_conv_fwd_grad(int(kind), x.data_ptr(), y.data_ptr(), v.data_ptr(), out.data_ptr(), gu.data_ptr(),
                                            B, N, M, D, float(blur), _dtype_code(x), *_range_args(ranges, B), *ws_args,
                                            int(flags), _stream(x))
    _check(rc, lib)
    return out, gu


def softmin_dense_fwd_raw(C, h, eps):
    lib = load_library()
    B, N, M = C.shape
    out = torch.empty((B, N), dtype=torch.float32, device=C.device)
    with torch.cuda.device(C.device):
        rc = lib.glhip_softmin_dense_fwd(C.data_ptr(), h.data_ptr(), out.data_ptr(), B, N, M, float(eps), _stream(C))
    _check(rc, lib)
    return out


def read_back(*tensors):
    """ONE host round trip for several small device tensors (cluster counts, kept-pair counts): their values as lists of ints."""
    same = all(t.dtype == tensors[0].dtype for t in tensors)      # (int32 cluster counts, int64 pair counts: no conversion launches)
    flat = torch.cat([t.reshape(-1) if same else t.reshape(-1).to(torch.int64) for t in tensors]).tolist()
    out, o = [], 0
    for t in tensors:
        out.append([int(v) for v in flat[o:o + t.numel()]])
        o += t.numel()
    return out


def grid_cluster_raw(x, weights, voxel, pre_div=1.0, gather=True, defer=False):
    """Voxel clustering on the device (``glhip_grid_cluster``).  x (N,D) fp32|bf16 contiguous CUDA, weights (N,) fp32 or None.

    Returns ``(perm, x_sorted, w_sorted, ranges, centroids, weights_c)``: ``perm`` (N,) int32; ``x_sorted`` / ``w_sorted`` the
    cloud in cluster order (None without ``gather``); ``ranges`` (C,2) int32, ``centroids`` (C,D) fp32 of ``x / pre_div``,
    ``weights_c`` (C,) fp32.  One host read-back (the cluster count) sizes the outputs — with ``defer`` the launch is queued and
    ``(count tensor, finish)`` comes back instead: the caller reads the counts of several clusterings in one round trip
    (:func:`read_back`) and calls ``finish(count values)`` for the tuple above.  The count tensor holds 8 integers, ``{C, overflow,
    qmin[3], qmax[3]}``: the voxel bounds of the cloud travel in the same round trip (:func:`voxel_extent`)."""
    lib = load_library()
    N, D = x.shape
    dev = x.device
    with torch.cuda.device(dev):
        perm = torch.empty(N, dtype=torch.int32, device=dev)
        x_sorted = torch.empty_like(x) if gather else None
        w_sorted = torch.empty(N, dtype=torch.float32, device=dev) if gather else None
        ranges = torch.empty((N, 2), dtype=torch.int32, device=dev)
        cents = torch.empty((N, D), dtype=torch.float32, device=dev)
        w_c = torch.empty(N, dtype=torch.float32, device=dev)
        count = torch.empty(8, dtype=torch.int32, device=dev)        # {C, overflow, qmin[3], qmax[3]}
        nbytes = int(lib.glhip_cluster_workspace_bytes(N, D))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        rc = lib.glhip_grid_cluster(x.data_ptr(), ptr(weights), N, D, _dtype_code(x), float(pre_div), float(voxel), perm.data_ptr(),
                                    ptr(x_sorted), ptr(w_sorted), ranges.data_ptr(), cents.data_ptr(), w_c.data_ptr(),
                                    count.data_ptr(), ws.data_ptr(), nbytes, _stream(x))
    _check(rc, lib)

    def finish(values):
        C, overflow = values[:2]
        if overflow:
            raise ValueError("geomloss_amd: the voxel grid has more than 2^21 cells along an axis; use a larger cluster_scale.")
        return perm, x_sorted, w_sorted, ranges[:C], cents[:C], w_c[:C]
    if defer:
        return count, finish
    return finish(read_back(count)[0])      # the one host round trip


def voxel_extent(count_values, voxel, pre_div=1.0):
    """Upper bound of the diagonal of the joint bounding box of the clouds whose ``glhip_grid_cluster`` count values (8 integers
    each, same ``voxel`` and ``pre_div``) are given: every cloud lies in [qmin, qmax + 1) voxels along each axis."""
    live = [v for v in count_values if v[0] > 0]
    if not live:
        return 0.0
    side = [(max(v[5 + d] for v in live) + 1 - min(v[2 + d] for v in live)) * float(voxel) * float(pre_div) for d in range(3)]
    return float(sum(s * s for s in side) ** 0.5)


# intervals: up to here the worst-case buffers (2 x 32 MB of address space the kernels touch the used part of) are cheaper than the
# counting pass (two more launches of 25 us per pattern) and its host round trip.  The ~2000 x 2000 clusters of the reference's
# voxel rule need 2e6: with 2^20 every pattern of a two-scale loss was counted first — three synchronisations and 0.3 ms of a
# 6.2-ms loss at N = 1e5 (round 5).
_RANGES_WORST_CASE_MAX = 1 << 22


def _worst_case_intervals(dev):
    """Largest worst-case interval count served without the counting pass: 2^22 = 2 x 32 MB of address space per pattern, of which
    the kernels touch the used part.  (A cap relative to the device's memory was tried after the round-5 advice:
    ``torch.cuda.get_device_properties`` costs ~110 ms on its first call — the whole first-call budget of a two-scale loss — and
    ``mem_get_info`` a driver round trip per pattern; the buffers are transient and come from torch's caching allocator.)"""
    return _RANGES_WORST_CASE_MAX


def block_ranges_raw(kind, rows, cols, f, g, ranges_rows, ranges_cols, thr, p=2, symmetric=False):
    """Keep rule on cluster pairs -> :class:`BlockRanges` (``glhip_block_ranges``).

    The interval buffers are sized for the worst case (every other cluster kept: Cr * ceil(Cc / 2)) while that is small —
    no host round trip — and from the counting pass (``glhip_block_ranges_count``, one read-back of two integers) beyond:
    the worst case is quadratic in the number of clusters, the real count is not."""
    lib = load_library()
    Cr, D = rows.shape
    Cc = cols.shape[0]
    dev = rows.device
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with torch.cuda.device(dev):
        # symmetric (rows is cols, f is g: the caller's word): the transposed pattern is the same arrays — one orientation is built
        symmetric = bool(symmetric) and Cr == Cc
        slices_r = torch.empty(Cr, dtype=torch.int32, device=dev)
        slices_c = None if symmetric else torch.empty(Cc, dtype=torch.int32, device=dev)
        head = (int(kind), rows.data_ptr(), cols.data_ptr(), ptr(f), ptr(g), Cr, Cc, D, int(p), float(thr),
                ranges_rows.data_ptr(), ranges_cols.data_ptr())
        cap = max(Cr * ((Cc + 1) // 2), Cc * ((Cr + 1) // 2), 1)
        if cap > _worst_case_intervals(dev):
            totals = torch.empty(2, dtype=torch.int32, device=dev)
            _check(lib.glhip_block_ranges_count(*head, slices_r.data_ptr(), (slices_r if symmetric else slices_c).data_ptr(), totals.data_ptr(),
                                                _stream(rows)), lib)
            counts = [int(v) for v in totals.tolist()]                   # the one host round trip of the big case
            if min(counts) < 0:      # int32 totals: a kept pattern of >= 2^31 intervals wraps around
                raise ValueError("geomloss_amd: the block-sparse pattern has more than 2^31 column intervals; use a larger cluster_scale.")
            cap = max(max(counts), 1)
            counted = True
        else:
            counted = False
        red_c = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        red_r = None if symmetric else torch.empty((cap, 2), dtype=torch.int32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)          # set by the kernel if an interval did not fit `cap`
        rc = lib.glhip_block_ranges(*head, slices_r.data_ptr(), red_c.data_ptr(), ptr(slices_c), ptr(red_r), cap,
                                    status.data_ptr(), _stream(rows))
    _check(rc, lib)
    # worst-case buffers cannot overflow; counted ones could only if the two passes disagreed: the host is already in step with
    # the stream there (it read the totals), so the check costs one more small read-back and nothing is dropped silently
    if counted and int(status.item()) != 0:
        raise RuntimeError("geomloss_amd: glhip_block_ranges wrote more intervals than its counting pass announced.")
    if symmetric:
        return BlockRanges(ranges_rows, slices_r, red_c, ranges_rows, slices_r, red_c)
    return BlockRanges(ranges_rows, slices_r, red_c, ranges_cols, slices_c, red_r)


# ----------------------------------------------------------------------------------------------
#  the elementwise front and back end of a small Sinkhorn loss as one launch each (glhip_log_weights, glhip_sinkhorn_cost)
# ----------------------------------------------------------------------------------------------

_SMALL_ENDS_MAX = 32768      # points per measure up to which the one-workgroup-per-item cost kernel beats torch's tree of reductions


def small_ends_apply(*tensors):
    """fp32 CUDA vectors of at most _SMALL_ENDS_MAX entries per batch item, library present: the regime where a loss is bound by the
    host's launch rate (a 2000-point loss: 26 launches, 9 of them soft-mins)."""
    return (library_available() and all(t is None or (t.is_cuda and t.dtype == torch.float32 and t.shape[-1] <= _SMALL_ENDS_MAX)
                                        for t in tensors))


_BBOX_MAX_POINTS = 16384      # one workgroup: beyond, the latency of its serial sweep exceeds the two torch reductions it replaces


def bounding_box_applies(x, y):
    """(n, D) contiguous fp32 / bf16 CUDA clouds of one dtype, D <= 16, at most _BBOX_MAX_POINTS points together."""
    return (library_available() and x.is_cuda and y.is_cuda and x.device == y.device and x.dtype == y.dtype
            and x.dtype in (torch.float32, torch.bfloat16) and x.dim() == 2 and y.dim() == 2 and x.shape[1] == y.shape[1] <= 16
            and x.is_contiguous() and y.is_contiguous() and 0 < x.shape[0] + y.shape[0] <= _BBOX_MAX_POINTS and _SMALL_ENDS_MAX > 0)


def bounding_box(x, y):
    """(mins, maxs) of the coordinates of x and y together, fp32 (D,) each: ``glhip_bounding_box``, one launch."""
    lib = load_library()
    D = x.shape[1]
    out = torch.empty(2 * D, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.glhip_bounding_box(x.data_ptr(), x.shape[0], y.data_ptr(), y.shape[0], D, _dtype_code(x), out.data_ptr(), _stream(x)), lib)
    return out[:D], out[D:]


def log_weights_raw(ws):
    """[log(w) with log(0) -> -100000 for w in ws] (up to 4 fp32 CUDA vectors of any shape) in ONE launch."""
    lib = load_library()
    ws = [w.detach().contiguous() for w in ws]
    outs = [torch.empty_like(w) for w in ws]
    k = len(ws)
    arr_w = (ctypes.c_void_p * k)(*[w.data_ptr() for w in ws])
    arr_o = (ctypes.c_void_p * k)(*[o.data_ptr() for o in outs])
    arr_n = (ctypes.c_long * k)(*[w.numel() for w in ws])
    with torch.cuda.device(ws[0].device):
        _check(lib.glhip_log_weights(arr_w, arr_o, arr_n, k, _stream(ws[0])), lib)
    return outs


def _cost_raw(a, f_ba, f_aa, b, g_ab, g_bb, B, N, M):
    lib = load_library()
    c = lambda t: None if t is None else t.detach().contiguous()  # noqa: E731
    a, f_ba, f_aa, b, g_ab, g_bb = (c(t) for t in (a, f_ba, f_aa, b, g_ab, g_bb))
    out = torch.empty(B, dtype=torch.float32, device=f_ba.device)
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with torch.cuda.device(f_ba.device):
        _check(lib.glhip_sinkhorn_cost(ptr(a), ptr(f_ba), ptr(f_aa), ptr(b), ptr(g_ab), ptr(g_bb), out.data_ptr(), B, N, M,
                                       int(a.numel() == B * N), int(b.numel() == B * M), _stream(f_ba)), lib)
    return out


class _SinkhornCost(torch.autograd.Function):
    """<a, f_ba - f_aa> + <b, g_ab - g_bb> per batch item as one launch (``sinkhorn_cost``, balanced; f_aa = g_bb = None: no
    debiasing).  The backward pass is written with differentiable torch operations of the inputs — the loss formula is bilinear —
    so that gradients of any order flow as they do through the seven torch operations it replaces."""

    @staticmethod
    def forward(ctx, batch, a, f_ba, f_aa, b, g_ab, g_bb):
        B = f_ba.shape[0] if batch else 1
        N, M = f_ba.shape[-1], g_ab.shape[-1]
        ctx.batch = batch
        ctx.save_for_backward(a, f_ba, f_aa, b, g_ab, g_bb)
        out = _cost_raw(a, f_ba, f_aa, b, g_ab, g_bb, B, N, M)
        return out if batch else out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, f_ba, f_aa, b, g_ab, g_bb = ctx.saved_tensors
        gg = g.reshape(-1, 1) if ctx.batch else g
        need = ctx.needs_input_grad
        like = lambda v, t: v.reshape(t.shape) if v.numel() == t.numel() else v.sum(0).reshape(t.shape)  # noqa: E731  (weights shared by the batch)
        ga = like(gg * (f_ba - f_aa if f_aa is not None else f_ba), a) if need[1] else None
        gb = like(gg * (g_ab - g_bb if g_bb is not None else g_ab), b) if need[4] else None
        wa = (gg * a).expand_as(f_ba) if (need[2] or need[3]) else None
        wb = (gg * b).expand_as(g_ab) if (need[5] or need[6]) else None
        return (None, ga, wa if need[2] else None, (-wa if (need[3] and f_aa is not None) else None), gb, wb if need[5] else None,
                (-wb if (need[6] and g_bb is not None) else None))


def sinkhorn_cost_fused(a, f_ba, f_aa, b, g_ab, g_bb, batch):
    return _SinkhornCost.apply(bool(batch), a, f_ba, f_aa, b, g_ab, g_bb)


def kept_pairs_raw(kind, rows, cols, f, g, ranges_rows, ranges_cols, thr, p=2, defer=False):
    """``(kept pairs of points, sum of squared row-cluster sizes, sum of squared column-cluster sizes)`` for the keep rule of
    :func:`block_ranges_raw` (``glhip_block_ranges_kept_pairs``; same arguments): one launch and one 24-byte read-back, no intervals
    built.  ``defer``: the (3,) int64 device tensor instead, for a shared :func:`read_back`."""
    lib = load_library()
    Cr, D = rows.shape
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with torch.cuda.device(rows.device):
        kept = torch.empty(3, dtype=torch.int64, device=rows.device)
        _check(lib.glhip_block_ranges_kept_pairs(int(kind), rows.data_ptr(), cols.data_ptr(), ptr(f), ptr(g), Cr, cols.shape[0], D, int(p),
                                                 float(thr), ranges_rows.data_ptr(), ranges_cols.data_ptr(), kept.data_ptr(),
                                                 _stream(rows)), lib)
    return kept if defer else tuple(read_back(kept)[0])


# ----------------------------------------------------------------------------------------------
#  distance-type reductions on the matrix cores (GLHIP_FLAG_MFMA_DIST): compact row blocks for dense launches
# ----------------------------------------------------------------------------------------------

# p = 1 soft-min / laplacian / energy: large dense launches are voxel-sorted first, so that every row block of the launch is
# spatially compact — the condition under which the squared distance may come from the MFMA (glhip_dist_x32.h)
_dist_on_mfma = os.environ.get("GEOMLOSS_HIP_MFMA_DIST", "1") != "0"
_DIST_MIN_ROWS, _DIST_MIN_PAIRS, _DIST_ROWS_PER_VOXEL, _DIST_COL_CHUNKS = 65536, 5e8, 256, 8
_DIST_SLAB = 256          # rows per row block of a compact-rows plan = the row tile of the distance kernel (8 wavefronts x 32 rows)


def set_distance_on_mfma(enabled):
    global _dist_on_mfma
    _dist_on_mfma = bool(enabled)


def _voxel_for(x, rows_per_voxel):
    """Voxel edge that puts ~rows_per_voxel points in an occupied voxel, from the bounding box of the (N, D) cloud."""
    N, D = x.shape
    # reduce along the contiguous axis of a (D, N) copy: torch's reduction over dim 0 of a row-major (N, 3) tensor runs at 25 GB/s
    # (0.8 ms for the two of them at 1e6 points, as long as the voxel sort they prepare)
    lo, hi = torch.aminmax(x.detach().float().t().contiguous(), dim=1)
    ext = [float(e) for e in (hi - lo).tolist()]
    live = [e for e in ext if e > 1e-6 * max(max(ext), 1e-30)]                     # axes the cloud really extends along
    vol = 1.0
    for e in live:
        vol *= e
    # never more than 2^20 voxels along an axis
    return max((vol * rows_per_voxel / N) ** (1.0 / max(len(live), 1)), max(ext) / (1 << 20), 1e-30)


def _serpentine(perm, xs, ranges, cents, voxel):
    """Re-orders the voxel clusters of a lexicographically voxel-sorted cloud along a boustrophedon path (the scan direction of
    every axis flips each time the path index of the axes before it advances), so that voxels that follow each other in memory
    are always face neighbours in space.  ANY run of consecutive rows of the result is then spatially compact — which is what
    lets the callers cut the rows into equal slabs instead of one (unevenly filled) block per voxel.
    perm (N,) int64, xs (N,D) sorted cloud, ranges (C,2), cents (C,D) -> (perm, xs) in the new order."""
    q = torch.floor(cents.float() / voxel).long()
    q = q - q.amin(0)
    ext = q.amax(0) + 1
    path = torch.zeros_like(q[:, 0])
    for a in range(q.shape[1]):
        c = torch.where(path % 2 == 1, ext[a] - 1 - q[:, a], q[:, a])
        path = path * ext[a] + c
    order = torch.argsort(path)
    sizes = (ranges[:, 1] - ranges[:, 0]).long()[order]
    starts = ranges[:, 0].long()[order]
    offs = torch.cumsum(sizes, 0) - sizes                                   # first row of every cluster in the new order
    # output_size: spares repeat_interleave the host round trip it would otherwise make to learn sum(sizes)
    idx = torch.repeat_interleave(starts - offs, sizes, output_size=xs.shape[0]) + torch.arange(xs.shape[0], device=xs.device)
    return perm[idx], xs[idx]


def compact_order(x, rows_per_voxel=None):
    """(perm, x[perm]): the cloud x (N,D) voxel-sorted (``glhip_grid_cluster``, ~rows_per_voxel points per voxel) with the voxels
    chained along the boustrophedon path of :func:`_serpentine` — any run of consecutive rows is spatially compact."""
    voxel = _voxel_for(x, _DIST_ROWS_PER_VOXEL if rows_per_voxel is None else rows_per_voxel)
    perm, xs, _, ranges, cents, _ = grid_cluster_raw(x.contiguous(), None, voxel)
    return _serpentine(perm.long(), xs, ranges, cents, voxel)


def autosort_applies(x, y, ranges=None, flags=0):
    """Whether a dense distance-type launch (p = 1 soft-min / half-step, laplacian / energy product) over these clouds sorts them
    inside the library (``csrc/glhip_autosort.h``: voxel sort along a boustrophedon path into the workspace, distances on the matrix
    cores over slabs of 256 compact rows, results back in the caller's order).  Until round 5 this was a Python-side plan
    (``_CompactRows``); it lives behind the C-ABI now, so a caller of ``glhip_softmin_fwd(p=1)`` gets it too.  Shapes and flags
    only: no device work."""
    B = 1 if x.dim() == 2 else x.shape[0]
    N, M, D = x.shape[-2], y.shape[-2], x.shape[-1]
    flags = int(flags) | ENV_FLAGS
    return not (not _dist_on_mfma or ranges is not None or B != 1 or D > 3 or N < _DIST_MIN_ROWS or is_f64(x)
                or float(N) * M < _DIST_MIN_PAIRS or (flags & (FLAG_NO_MFMA | FLAG_DIRECT | FLAG_NO_SORT)))


# ----------------------------------------------------------------------------------------------
#  autograd functions
# ----------------------------------------------------------------------------------------------

def _plan_moments(xb, yb, hb, out, eps, ranges, flags):
    """S_i = sum_j P_ij u_j u_j^T (B,N,D,D) and the centre c (B,1,D), u_j = y_j - c, for the plan P_ij = exp(h_j - |x_i - y_j|^2 / (2 eps)
    + out_i / eps) of a p = 2 soft-min — from the EXISTING gradient reduction (``glhip_softmin_bwd_x``), on augmented clouds:
    x' = (x, 0), y' = (y, t q_j), h' = h + t^2 |q_j|^2 / (2 eps) leave every P_ij unchanged (the extra squared distance cancels
    against the shift of h), and the extra components of sum_j P_ij (x'_i - y'_j) are -t sum_j P_ij q_j: any average of column
    features q_j under the plan, here q = the D (D + 1) / 2 products u_a u_b.  t = sqrt(eps) / max |q| keeps the shift of h of
    order one; D + features <= 16 per call keeps the matrix-core kernels."""
    B, N, D = xb.shape
    xf, yf = xb.float(), yb.float()
    c = yf.mean(1, keepdim=True)
    u = yf - c
    ia, ib = torch.triu_indices(D, D, device=xb.device)
    q = u[..., ia] * u[..., ib]                                    # (B, M, K)
    K = q.shape[-1]
    t = (eps ** 0.5) / q.abs().amax().clamp_min(1e-30)             # device scalar: no host round trip
    chunk = (XD_MAX_DIM - D) if D <= XD_MAX_DIM - 4 else 8
    ones = torch.ones((B, N), dtype=torch.float32, device=xb.device)
    mom = torch.empty((B, N, K), dtype=torch.float32, device=xb.device)
    flags = int(flags) & ~(FLAG_F16X2 | FLAG_MFMA_DIST)
    for k0 in range(0, K, chunk):
        qa = q[..., k0:k0 + chunk] * t
        ya = torch.cat([yf, qa], -1).contiguous()
        xa = torch.cat([xf, torch.zeros((B, N, qa.shape[-1]), dtype=torch.float32, device=xb.device)], -1).contiguous()
        ha = (hb.float() + (qa * qa).sum(-1) / (2.0 * eps)).contiguous()
        r = softmin_bwd_x_raw(xa, ya, ha, out, ones, eps, 2, ranges, flags)
        mom[..., k0:k0 + chunk] = r[..., D:] / (-t)
    S = torch.empty((B, N, D, D), dtype=torch.float32, device=xb.device)
    S[..., ia, ib] = mom
    S[..., ib, ia] = mom
    return S, c


class _SoftminBwdX(torch.autograd.Function):
    """G_i = g_i d f_i / d x_i, the backward pass of the soft-min as a differentiable operation of (x, g): what
    ``torch.autograd.grad(..., create_graph=True)`` records (KeOps' `Grad` of `_legacy/sinkhorn_samples.py:322-334`
    is differentiable again, `_legacy/sinkhorn_divergence.py:612-623` leaves autograd on for that).  y and h carry no gradient:
    the Sinkhorn loop detaches them.

    p = 2: d f_i / d x_i = x_i - ybar_i with ybar_i = sum_j P_ij y_j, so for a cotangent V (B,N,D)
        d<V, G> / d g_i = V_i . (x_i - ybar_i)
        d<V, G> / d x_i = g_i [ V_i - Cov_i V_i / eps ],   Cov_i = sum_j P_ij y_j y_j^T - ybar_i ybar_i^T
    (d P_ij / d x_i = P_ij (y_j - ybar_i) / eps).  The second moments come from :func:`_plan_moments`: one or a few more calls of
    the gradient kernel, O(N + M) memory at any size.  p = 1: the Hessian of |x - y| makes the column features depend on the row;
    d / d g is served, d / d x raises (``backend="tensorized"`` is differentiable to any order)."""

    @staticmethod
    def forward(ctx, x, g, yb, hb, out, eps, p, ranges, flags):
        xb = _points(x.detach(), "x", True)
        ones = torch.ones_like(out)
        unit = softmin_bwd_x_raw(xb, yb, hb, out, ones, eps, p, ranges, flags)       # d f_i / d x_i
        ctx.save_for_backward(xb, g.detach(), yb, hb, out, unit)
        ctx.cfg = (eps, p, ranges, flags, x.dtype, g.dtype)
        return (g.detach().to(unit.dtype).unsqueeze(-1) * unit).to(x.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, V):
        xb, g, yb, hb, out, unit = ctx.saved_tensors
        eps, p, ranges, flags, xdtype, gdtype = ctx.cfg
        V = V.reshape(unit.shape).to(unit.dtype)
        grad_g = (V * unit).sum(-1).to(gdtype) if ctx.needs_input_grad[1] else None
        grad_x = None
        if ctx.needs_input_grad[0]:
            if p != 2 or is_f64(xb):
                raise NotImplementedError(
                    "geomloss_amd: second-order derivatives with respect to the points are implemented for the p = 2 HIP soft-min "
                    "on float32 / bfloat16 clouds; use backend='tensorized' (differentiable to any order) for p = 1 or float64.")
            S, c = _plan_moments(xb, yb, hb, out, eps, ranges, flags)
            ubar = (xb.float() - c) - unit                                           # sum_j P_ij (y_j - c)
            cov_v = torch.einsum("bnde,bne->bnd", S, V) - ubar * (ubar * V).sum(-1, keepdim=True)
            grad_x = (g.to(unit.dtype).unsqueeze(-1) * (V - cov_v / eps)).to(xdtype)
        return grad_x, grad_g, None, None, None, None, None, None, None


def _bwd_x(x, g, yb, hb, out, eps, p, ranges, flags):
    """g_i d f_i / d x_i (B,N,D): one reduction — or, while autograd records the backward pass (``create_graph=True``), the
    differentiable node above.  ``x``: the (B,N,D) view of the tensor the caller differentiates."""
    if torch.is_grad_enabled() and (x.requires_grad or g.requires_grad):
        return _SoftminBwdX.apply(x, g, yb, hb, out, eps, p, ranges, flags)
    return softmin_bwd_x_raw(_points(x.detach(), "x", True), yb, hb, out, g.detach().to(out.dtype).contiguous(), eps, p, ranges, flags)


class _Softmin(torch.autograd.Function):
    """f_i = -eps log sum_j exp(h_j - C(x_i,y_j)/eps); differentiable in x only, like the reference's call sites."""

    @staticmethod
    def forward(ctx, x, y, h, eps, p, ranges, flags):
        xb, yb, hb, batched = _as_batched(_points(x, "x", True), _points(y, "y", True), h.detach().contiguous())
        if yb.dtype != xb.dtype:
            yb = yb.to(xb.dtype)
        if p == 1 and not is_f64(xb) and not (flags & (FLAG_NO_MFMA | FLAG_DIRECT)):
            if ranges is not None and _dist_on_mfma:
                flags |= FLAG_MFMA_DIST            # multiscale: the row blocks are voxel clusters already
            elif not _dist_on_mfma:
                flags |= FLAG_NO_SORT              # (big dense launches sort themselves inside the library unless told not to)
        out = softmin_fwd_raw(xb, yb, hb, eps, p, ranges, flags)
        ctx.save_for_backward(x, yb, hb, out)         # (x itself: under create_graph the backward pass is differentiated through it)
        ctx.cfg = (eps, p, ranges, flags, xb.shape)
        return out if batched else out.view(-1)

    @staticmethod
    def backward(ctx, grad_out):      # differentiable once more under create_graph=True (_SoftminBwdX)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise NotImplementedError(
                "geomloss_amd: the HIP soft-min is differentiable with respect to its first point cloud only "
                "(the Sinkhorn loop detaches the second cloud and the dual vector)."
            )
        x, yb, hb, out = ctx.saved_tensors
        eps, p, ranges, flags, bshape = ctx.cfg
        gx = _bwd_x(x.reshape(bshape), grad_out.reshape(out.shape), yb, hb, out, eps, p, ranges, flags)
        return gx.reshape(x.shape).to(x.dtype), None, None, None, None, None, None


class _SoftminValueGrad(torch.autograd.Function):
    """The soft-min together with its x-gradient from ONE reduction, for callers that can bound the answer (the last update of
    the Sinkhorn loop): forward runs ``glhip_softmin_fwd_grad`` and keeps d out_i / d x_i, backward is an elementwise product."""

    @staticmethod
    def forward(ctx, x, y, h, eps, guess, margin, ranges, flags):
        xb, yb, hb, batched = _as_batched(_points(x, "x"), _points(y, "y"), _f32(h))
        if yb.dtype != xb.dtype:
            yb = yb.to(xb.dtype)
        out, unit = softmin_fwd_grad_raw(xb, yb, hb, _f32(guess).reshape(hb.shape[0], -1), margin, eps, ranges, flags)
        ctx.unit, ctx.cfg = unit, (x.shape, x.dtype, eps, ranges, flags)
        ctx.save_for_backward(x, yb, hb, out)         # for a backward pass that is differentiated again (create_graph=True)
        return out if batched else out.view(-1)

    @staticmethod
    def backward(ctx, grad_out):
        xshape, xdtype, eps, ranges, flags = ctx.cfg
        g = grad_out.reshape(ctx.unit.shape[0], -1)
        x, yb, hb, out = ctx.saved_tensors
        if torch.is_grad_enabled() and (x.requires_grad or g.requires_grad):
            gx = _SoftminBwdX.apply(x.reshape(ctx.unit.shape), g, yb, hb, out, eps, 2, ranges, flags)
        else:
            gx = g.float().unsqueeze(-1) * ctx.unit
        return gx.reshape(xshape).to(xdtype), None, None, None, None, None, None, None


_VALUE_GRAD_MIN_PAIRS = 5e8
_VALUE_GRAD_MAX_MARGIN = 25.0       # in units of eps: the weights stay >= exp(-50) of the largest one


def softmin_value_and_grad(eps, x, y, h, guess, margin, ranges=None, flags=0):
    """``softmin(eps, x, y, h)`` (differentiable in x) from one reduction instead of a forward and a backward one, or None when that
    does not apply: x needs no gradient, p = 2 / D <= 16 kernels only, launches too small to pay for the host check of ``margin``
    (a device scalar: sup |h - h_previous| * eps), or a margin beyond 25 eps (the loop has not converged: use the two passes)."""
    if (not (torch.is_grad_enabled() and x.requires_grad) or x.shape[-1] > XD_MAX_DIM or is_f64(x)
            or (int(flags) | ENV_FLAGS) & (FLAG_NO_MFMA | FLAG_DIRECT)):
        return None
    rows, cols = x.shape[-2], y.shape[-2]
    B = 1 if x.dim() == 2 else x.shape[0]
    if float(B) * rows * cols < _VALUE_GRAD_MIN_PAIRS:
        return None
    m = float(margin)                       # the one host round trip of this path
    if not (m >= 0.0) or m > _VALUE_GRAD_MAX_MARGIN * eps:
        return None
    guess = torch.nan_to_num(guess.detach().float(), nan=0.0, posinf=0.0, neginf=0.0)     # rows without columns: any finite guess
    return _SoftminValueGrad.apply(x, y.detach(), h.detach(), float(eps), guess, m * 1.0001 + 1e-12, ranges, int(flags) | ENV_FLAGS)


def sinkhorn_iter4(eps, x, y, a_log, b_log, pots, damping, debias=True, flags=0, p=2):
    """One whole iteration of the symmetric Sinkhorn loop on the GPU, non-differentiable (dense, p = 1 or 2, D <= 16).

    x: (N,D)|(B,N,D), y: (M,D)|(B,M,D); a_log: (N,)|(1,N)|(B,N), b_log likewise; ``pots`` None (initial potentials)
    or the old ``(f_ba, g_ab, f_aa, g_bb)`` / ``(f_ba, g_ab)``.  Returns the new potentials, shaped like a_log / b_log."""
    xb, yb, bl, _ = _as_batched(_points(x.detach(), "x"), _points(y.detach(), "y"), _f32(b_log))
    if yb.dtype != xb.dtype:
        yb = yb.to(xb.dtype)
    B = xb.shape[0]
    al = _f32(a_log).reshape(B, -1)
    old = None if pots is None else tuple(_f32(t).reshape(B, -1) for t in pots)
    new = sinkhorn_iter4_raw(xb, yb, al, bl, old, eps, damping, debias, int(flags) | (ENV_FLAGS & (FLAG_NO_SPLIT | FLAG_F16X2)), p)
    shapes = (a_log.shape, b_log.shape, a_log.shape, b_log.shape)
    return tuple(t.view(sh) for t, sh in zip(new, shapes))


def sinkhorn_extrapolate4(eps, x, y, xc, yc, a_log_c, b_log_c, pots, damping, flags=0, p=2):
    """Coarse-to-fine jump of the two-scale loop on the GPU, non-differentiable: every potential of ``pots`` — (f_ba, g_ab, f_aa, g_bb)
    or (f_ba, g_ab), living on the coarse clouds xc (Nc,D), yc (Mc,D) — carried to the fine clouds x (N,D), y (M,D) by one soft-min
    against the coarse measure each, in ONE launch (dense, p = 1 or 2, D <= 16).  Batched (B,.,D) clouds alike."""
    pts = [_points(t.detach(), n) for t, n in ((x, "x"), (y, "y"), (xc, "xc"), (yc, "yc"))]
    if any(t.dtype != pts[0].dtype for t in pts):
        pts = [t.to(pts[0].dtype) for t in pts]
    batched = pts[0].dim() == 3
    xb, yb, xcb, ycb = (t if batched else t[None] for t in pts)
    B = xb.shape[0]
    row = lambda t: _f32(t).reshape(B, -1)      # noqa: E731
    outs = sinkhorn_extrapolate4_raw(xb, yb, xcb, ycb, row(a_log_c), row(b_log_c), tuple(row(t) for t in pots), eps, damping,
                                     int(flags) | (ENV_FLAGS & (FLAG_NO_SPLIT | FLAG_F16X2)), p)
    return outs if batched else tuple(t.view(-1) for t in outs)


class Iter4Plan:
    """Everything about a run of ``glhip_sinkhorn_iter4`` calls that does not change from one iteration to the next:
    validated contiguous inputs, the scratch buffer, two sets of output buffers (an iteration reads the potentials the
    previous one wrote, and outputs may not alias inputs).  At N ~ 1e3 the per-call Python work (checks, allocations,
    device guard) costs more than the kernels; the plan leaves one ctypes call per iteration."""

    def __init__(self, x, y, a_log, b_log, debias=True, flags=0, p=2):
        self.lib = load_library()
        self.p = int(p)
        xb, yb, bl, _ = _as_batched(_points(x.detach(), "x"), _points(y.detach(), "y"), _f32(b_log))
        if yb.dtype != xb.dtype:
            yb = yb.to(xb.dtype)
        self.x, self.y, self.b_log = xb, yb, bl
        B, N, D = xb.shape
        M = yb.shape[1]
        self.a_log = _f32(a_log).reshape(B, -1)
        self.dims = (B, N, M, D)
        self.debias = debias
        self.flags = int(flags) | (ENV_FLAGS & (FLAG_NO_SPLIT | FLAG_F16X2))
        self.shapes = (a_log.shape, b_log.shape, a_log.shape, b_log.shape)[: 4 if debias else 2]
        sizes = (N, M, N, M)[: 4 if debias else 2]
        dev = xb.device
        self.device = dev
        with torch.cuda.device(dev):
            L = max(N, M)
            self.nbytes = 4 * int(self.lib.glhip_workspace_bytes(B, L, L, D, 0))
            self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=dev) if self.nbytes else None
            self.sets = []
            for _ in range(2):
                flat = torch.empty(B * sum(sizes), dtype=torch.float32, device=dev)
                bufs, o = [], 0
                for n in sizes:
                    bufs.append(flat[o:o + B * n].view(B, n))
                    o += B * n
                self.sets.append(bufs)
        self.shaped = [tuple(t.view(sh) for t, sh in zip(bufs, self.shapes)) for bufs in self.sets]     # what run / anneal hand out
        self.turn = 0
        self.extra_flags = 0      # per-call flags of the owner (GLHIP_FLAG_F16X2 where the temperature allows: sinkhorn_samples._HipSoftmin)
        self.fixed = (xb.data_ptr(), yb.data_ptr(), self.a_log.data_ptr(), bl.data_ptr())
        self.dtype = _dtype_code(xb)

    def run(self, eps, damping, pots, last=False):
        """pots None: initial potentials.  Otherwise the averaged update, or with ``last`` the plain, non-averaged
        update ``damping * softmin(eps, C, logw + pot/eps)`` written to fresh tensors (the differentiable step)."""
        B, N, M, D = self.dims
        if last:      # fresh tensors, in the shapes of the log-weights (same element counts as (B, N) / (B, M))
            outs = shaped = tuple(torch.empty(sh, dtype=torch.float32, device=self.device) for sh in self.shapes)
        else:
            outs, shaped = self.sets[self.turn], self.shaped[self.turn]
            self.turn ^= 1
        if pots is None:
            old = (None, None, None, None)
        else:
            old = tuple(t.data_ptr() if (t.dtype is torch.float32 and t.is_contiguous()) else None for t in pots)
            if None in old:   # unusual inputs: normalise (keeps them alive until the launch is queued)
                pots = tuple(_f32(t).reshape(B, -1) for t in pots)
                old = tuple(t.data_ptr() for t in pots)
            old = old + (None,) * (4 - len(old))
        new = tuple(t.data_ptr() for t in outs) + (None,) * (4 - len(outs))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.glhip_sinkhorn_iter4(*self.fixed, *old, *new, B, N, M, D, float(eps), float(damping), self.p, self.dtype,
                                           1 if pots is None else (2 if last else 0), None if self.ws is None else self.ws.data_ptr(), self.nbytes,
                                           self.flags | self.extra_flags, stream)
        _check(rc, self.lib)
        return shaped

    def anneal(self, eps_list, dampings, f16x2_min_eps=float("inf")):
        """The initialisation at ``eps_list[0]`` and one averaged iteration per temperature, queued by ONE library call
        (``glhip_sinkhorn_anneal``) into the plan's two buffer sets.  Returns ``(final potentials, inputs of the last iteration)``,
        both shaped like the log-weights; the next :meth:`run` writes over the latter.  GLHIP_FLAG_F16X2 (with the plan's other
        flags) applies to the temperatures >= ``f16x2_min_eps``."""
        B, N, M, D = self.dims
        n = len(eps_list)
        ptrs = ctypes.c_void_p * 4
        sets = [ptrs(*([t.data_ptr() for t in bufs] + [None] * (4 - len(bufs)))) for bufs in self.sets]
        farr = ctypes.c_float * n
        stream = torch.cuda.current_stream(self.device).cuda_stream
        flags, min_eps = self.flags, min(float(f16x2_min_eps), 3.0e38)
        if flags & FLAG_F16X2:           # forced for every launch through the environment (GEOMLOSS_HIP_FLAGS)
            min_eps = 0.0
        elif min_eps < 3.0e38:
            flags |= FLAG_F16X2
        rc = self.lib.glhip_sinkhorn_anneal(*self.fixed, sets[0], sets[1], B, N, M, D, farr(*[float(e) for e in eps_list]),
                                            farr(*[float(d) for d in dampings]), n, self.p, self.dtype,
                                            None if self.ws is None else self.ws.data_ptr(), self.nbytes, flags, min_eps, stream)
        _check(rc, self.lib)
        self.turn = (n + 1) % 2          # the set the final potentials are NOT in
        return self.shaped[n % 2], self.shaped[(n + 1) % 2]


class _Last4(torch.autograd.Function):
    """The non-averaged last update of the loop (sinkhorn_divergence.py:612-623) for all potentials at once: one forward
    launch (``glhip_sinkhorn_iter4``, first = 2); differentiable in the row cloud of each cost (x for f_ba / f_aa, y for
    g_ab / g_bb), exactly like the four separate soft-mins it replaces."""

    @staticmethod
    def forward(ctx, x, y, plan, eps, damping, *pots):
        outs = plan.run(eps, damping, tuple(p.detach() for p in pots), last=True)
        ctx.plan, ctx.cfg, ctx.extra_flags = plan, (eps, damping, x.shape, x.dtype, y.shape, y.dtype), plan.extra_flags
        ctx.save_for_backward(x, y, *(p.detach() for p in pots), *outs)
        return outs

    @staticmethod
    def backward(ctx, *grads):        # differentiable once more under create_graph=True (_SoftminBwdX)
        plan = ctx.plan
        eps, damping, xshape, xdtype, yshape, ydtype = ctx.cfg
        B = plan.dims[0]
        x_in, y_in = ctx.saved_tensors[:2]
        saved = ctx.saved_tensors[2:]
        k = len(saved) // 2
        pots, outs = saved[:k], saved[k:]
        # (rows, columns, log-weights of the columns, potential carried by the columns) of each reduction
        specs = [(x_in, plan.y, plan.b_log, pots[1]), (y_in, plan.x, plan.a_log, pots[0])]
        if k == 4:
            specs += [(x_in, plan.x, plan.a_log, pots[2]), (y_in, plan.y, plan.b_log, pots[3])]
        gx = gy = None
        for i, (rows, cols, logw, pot) in enumerate(specs):
            g = grads[i]
            if g is None or not ctx.needs_input_grad[i % 2]:
                continue
            with torch.no_grad():      # (only for the reductions that take a gradient: a loss differentiated in x alone uses 2 of the 4 —
                # the other two cost 6 launches of a launch-bound backward pass, round 6)
                h_i = logw + pot.reshape(B, -1) * (1.0 / eps)
                val_i = (outs[i].reshape(B, -1) * (1.0 / damping)).contiguous()      # the soft-min values themselves
            gr = _bwd_x(rows.reshape(B, -1, rows.shape[-1]), g.reshape(B, -1) * damping, cols, h_i, val_i, eps, plan.p, None,
                        plan.flags | ctx.extra_flags).float()
            if i % 2 == 0:
                gx = gr if gx is None else gx + gr
            else:
                gy = gr if gy is None else gy + gr
        gx = None if gx is None else gx.reshape(xshape).to(xdtype)
        gy = None if gy is None else gy.reshape(yshape).to(ydtype)
        return (gx, gy, None, None, None) + (None,) * k


def sinkhorn_last4(plan, x, y, eps, damping, pots):
    """Differentiable last update through an :class:`Iter4Plan`; returns the new potentials shaped like the old ones."""
    if not (torch.is_grad_enabled() and (x.requires_grad or y.requires_grad)):      # nothing to differentiate: no autograd node
        return plan.run(float(eps), float(damping), tuple(p.detach() for p in pots), last=True)
    return _Last4.apply(x, y, plan, float(eps), float(damping), *pots)        # (Iter4Plan.run hands them out in those shapes)


# kernel-selection knobs for A/B runs (tuning through the environment only): a GLHIP_FLAG_* bitmask
ENV_FLAGS = int(os.environ.get("GEOMLOSS_HIP_FLAGS", "0"))


def softmin(eps, x, y, h, p=2, ranges=None, flags=0):
    """Soft-C-transform on the GPU.  x: (N,D)|(B,N,D), y: (M,D)|(B,M,D), h: (M,)|(B,M) -> (N,)|(B,N) fp32."""
    return _Softmin.apply(x, y, h, float(eps), int(p), ranges, int(flags) | ENV_FLAGS)


def fused_step_applies(D, p=2, flags=0, sparse=False):
    """Whether ``glhip_sinkhorn_step`` has a kernel for clouds of dimension D: every operator for D <= 3; for 4 <= D <= XD_MAX_DIM
    the matrix-core kernels — p = 2 (glhip_softmin_xd.h), and since round 5 p = 1 on DENSE launches (glhip_dist_xd.h; ``sparse``:
    the launch carries block-sparse ranges) — which GLHIP_FLAG_NO_MFMA / GLHIP_FLAG_DIRECT switch off: the generic-dimension
    kernel those flags fall back to has no fused half-step."""
    if D <= 3:
        return True
    if D > XD_MAX_DIM or ((int(flags) | ENV_FLAGS) & (FLAG_NO_MFMA | FLAG_DIRECT)):
        return False
    return p == 2 or (p == 1 and not sparse)


def sinkhorn_step(eps, x, y, logw, pot, prev, damping, p=2, ranges=None, flags=0):
    """One non-differentiable half-step of the Sinkhorn loop on the GPU: (prev + damping * softmin(eps, C, logw + pot/eps)) / 2,
    or damping * softmin(...) when prev is None — ONE launch where :func:`fused_step_applies`, the soft-min kernel followed by
    torch arithmetic elsewhere (D > 16, block-sparse p = 1 in D > 3, D > 3 under GLHIP_FLAG_NO_MFMA / GLHIP_FLAG_DIRECT).

    x: (N,D)|(B,N,D), y: (M,D)|(B,M,D); logw, pot: (M,)|(B,M) (pot may be None); prev: (N,)|(B,N) or None.
    Returns fp32 (N,)|(B,N).  Used by the drivers inside the no-grad part of ``sinkhorn_loop``."""
    if is_f64(x) and p in (1, 2):      # float64 clouds: the fused double-precision half-step (any D, dense / batched / block-sparse)
        xb, yb, lw, batched = _as_batched(_points(x.detach(), "x", allow_f64=True), _points(y.detach(), "y", allow_f64=True),
                                          logw.detach().double().contiguous())
        B = xb.shape[0]
        pt = None if pot is None else pot.detach().double().contiguous().reshape(B, -1)
        pv = None if prev is None else prev.detach().double().contiguous().reshape(B, -1)
        out = sinkhorn_step_raw(xb, yb.double(), lw, pt, pv, eps, damping, p, ranges, 0)
        return out if batched else out.view(-1)
    if not fused_step_applies(x.shape[-1], p, flags, ranges is not None) or is_f64(x):
        with torch.no_grad():
            h = _vec(logw, x) if pot is None else _vec(logw, x) + _vec(pot, x).reshape(logw.shape) / eps
            ft = damping * softmin(eps, x.detach(), y.detach(), h, p=p, ranges=ranges, flags=flags)
            return ft if prev is None else 0.5 * (_vec(prev, x).reshape(ft.shape) + ft)
    xb, yb, lw, batched = _as_batched(_points(x.detach(), "x"), _points(y.detach(), "y"), _f32(logw))
    if yb.dtype != xb.dtype:
        yb = yb.to(xb.dtype)
    B = xb.shape[0]
    pt = None if pot is None else _f32(pot).reshape(B, -1)
    pv = None if prev is None else _f32(prev).reshape(B, -1)
    flags = int(flags) | ENV_FLAGS
    if p == 1 and not (flags & (FLAG_NO_MFMA | FLAG_DIRECT)):
        if ranges is not None and _dist_on_mfma:
            flags |= FLAG_MFMA_DIST
        elif not _dist_on_mfma:
            flags |= FLAG_NO_SORT
    out = sinkhorn_step_raw(xb, yb, lw, pt, pv, eps, damping, p, ranges, flags)
    return out if batched else out.view(-1)


# Gaussian reductions on the kernels that centre every workgroup on its own first row (the gradient kernels and the products of
# their family, glhip_wsum_mfma.h): the exponent -|x - y|^2 / 2 blur^2 is assembled on the matrix cores from terms of size
# |x - c| |y - c| / blur^2, and its float32 accumulation leaves a relative error of that size x 2^-24 in the kernel value — a
# common bias of -1.6e-5 on every term of a norm at blur = .05 in the unit cube when the rows of a workgroup are scattered over
# the cloud.  With the rows in compact order (256 neighbours per workgroup) |x - c| is the size of a voxel and the bias drops to
# -1.2e-6 (profiles/r03_upper_triangle.txt).  Large dense launches only: the sort costs ~1 ms at 1e6 points.
_GAUSS_SORT_MIN_PAIRS = 1e11


def _gauss_compact_rows(kind, xb, M, ranges, flags):
    """(perm, rows in compact order) for a big dense gaussian launch of the per-workgroup-centre kernels, or None."""
    B, N, D = xb.shape
    if (kind != GAUSSIAN or ranges is not None or B != 1 or D > 3 or is_f64(xb) or float(N) * M < _GAUSS_SORT_MIN_PAIRS
            or (flags & (FLAG_NO_MFMA | FLAG_DIRECT))):
        return None
    perm, xs = compact_order(xb[0])
    return perm, xs.unsqueeze(0).contiguous()


def _unsort_rows(perm, t):
    """(1, N, ...) in sorted row order -> original order."""
    out = torch.empty_like(t)
    out[0, perm] = t[0]
    return out


class _KernelConv(torch.autograd.Function):
    """out_i = sum_j k(x_i,y_j) v_j, differentiable in x, y and v."""

    @staticmethod
    def forward(ctx, kind, x, y, v, blur, ranges, flags, x_grad=True):
        # x_grad: needs_input_grad says True for a leaf that requires gradients even under no_grad, where nothing will be asked for
        xb, yb, vb, batched, out, unit = _KernelConv.product(kind, x, y, v, blur, ranges, flags, x_grad and ctx.needs_input_grad[1])
        ctx.unit = unit
        ctx.save_for_backward(xb, yb, vb, x, y, v)      # x, y, v themselves: the differentiable backward (create_graph) needs their history
        ctx.cfg = (kind, blur, ranges, flags, x.shape, y.shape, v.shape, x.dtype, y.dtype, v.dtype)
        return out if batched else out.view(-1)

    @staticmethod
    def product(kind, x, y, v, blur, ranges, flags, want_unit):
        """The launch behind forward: (xb, yb, vb, batched, out (B,N), unit (B,N,D) | None); unit_i = d out_i / d x_i when
        ``want_unit`` and a product-and-gradient kernel exists for this kind and dimension."""
        xb, yb, vb, batched = _as_batched(_points(x, "x", True), _points(y, "y", True), v.detach().contiguous())
        if yb.dtype != xb.dtype:
            yb = yb.to(xb.dtype)
        if is_f64(xb):      # double precision: the plain product kernel, no plans, no fused gradient
            return xb, yb, vb, batched, kernel_conv_fwd_raw(kind, xb, yb, vb, blur, ranges, flags), None
        # When x requires gradients, the product and its row gradient come out of ONE reduction: the gradient kernel
        # carries one more accumulator, the product itself.  The backward pass is then elementwise.
        # (D <= 3: every kernel; 4 <= D <= 16: the gaussian kernel on the matrix cores)
        fused = (_fuse_kernel_grad and want_unit
                 and (xb.shape[-1] <= 3 or (kind == GAUSSIAN and xb.shape[-1] <= XD_MAX_DIM and not (flags & FLAG_NO_MFMA))))
        # laplacian / energy: squared distances from the matrix cores wherever the row blocks are spatially compact — the voxel
        # clusters of the multiscale backend as they are, large dense launches after the voxel sort the library does itself
        # (csrc/glhip_autosort.h) — for the product, the product + gradient and (GRAD_FAMILY) the companion products of a norm alike
        fl = flags
        if kind != GAUSSIAN and xb.shape[-1] <= 3 and not (flags & (FLAG_NO_MFMA | FLAG_DIRECT)):
            if ranges is not None and _dist_on_mfma:
                fl |= FLAG_MFMA_DIST
            elif not _dist_on_mfma:
                fl |= FLAG_NO_SORT
        rows = None
        X, Y, V, R = xb, yb, vb, ranges
        if fused or (flags & FLAG_GRAD_FAMILY):      # gaussian: the kernels with one centre per workgroup
            rows = _gauss_compact_rows(kind, xb, yb.shape[1], ranges, flags)
            if rows is not None:
                X = rows[1]
        if fused:
            out, unit = kernel_conv_fwd_grad_raw(kind, X, Y, V, blur, R, fl)
        else:
            out, unit = kernel_conv_fwd_raw(kind, X, Y, V, blur, R, fl), None
        if rows is not None:
            out, unit = _unsort_rows(rows[0], out), (None if unit is None else _unsort_rows(rows[0], unit))
        return xb, yb, vb, batched, out, unit

    @staticmethod
    def _differentiable_backward(ctx, grad_out):
        """backward under ``create_graph=True``: the three gradients written with differentiable operations, so that autograd can
        go through them once more (KeOps' symbolic ``Grad`` composes the same way: ``_legacy/kernel_samples.py:43-54``).  Gaussian
        kernel, dense: with s = 1 / blur^2 and k = exp(-s |x - y|^2 / 2),
            d/dx_i = -s g_i [ x_i (K v)_i - (K (v y))_i ],   d/dy_j = s v_j [ (K^T (g x))_j - y_j (K^T g)_j ],   d/dv = K^T g
        — D + 1 kernel products per cloud instead of one gradient reduction, each of them this very autograd function."""
        kind, blur, ranges, flags = ctx.cfg[:4]
        x, y, v = ctx.saved_tensors[3:]
        g = grad_out.reshape(x.shape[:-1])
        v = v.reshape(y.shape[:-1])
        s = 1.0 / (blur * blur)
        conv = lambda rows, cols, w: kernel_conv(kind, rows, cols, w, blur, None, flags & ~FLAG_GRAD_FAMILY)   # noqa: E731
        D = x.shape[-1]
        gx = gy = gv = None
        if ctx.needs_input_grad[1]:
            Kvy = torch.stack([conv(x, y, v * y[..., d]) for d in range(D)], dim=-1)
            gx = (-s) * g.unsqueeze(-1) * (x * conv(x, y, v).unsqueeze(-1) - Kvy)
        if ctx.needs_input_grad[2]:
            Kgx = torch.stack([conv(y, x, g * x[..., d]) for d in range(D)], dim=-1)
            gy = s * v.unsqueeze(-1) * (Kgx - y * conv(y, x, g).unsqueeze(-1))
        if ctx.needs_input_grad[3]:
            gv = conv(y, x, g)
        return None, gx, gy, gv, None, None, None, None

    @staticmethod
    def backward(ctx, grad_out):
        # create_graph=True: somebody may differentiate this gradient.  Dense gaussian products: the differentiable form; laplacian /
        # energy / block-sparse ones: the one-reduction gradients below, marked once-differentiable (autograd raises if a second
        # derivative is really taken through them; backend="tensorized" is differentiable to any order)
        if torch.is_grad_enabled() and ctx.cfg[0] == GAUSSIAN and ctx.cfg[2] is None:
            return _KernelConv._differentiable_backward(ctx, grad_out)
        return _KernelConv._backward_once(ctx, grad_out)

    @staticmethod
    @once_differentiable
    def _backward_once(ctx, grad_out):
        xb, yb, vb = ctx.saved_tensors[:3]
        kind, blur, ranges, flags, xs, ys, vs, xdt, ydt, vdt = ctx.cfg
        g = grad_out.reshape(xb.shape[0], -1).to(vb.dtype).contiguous()
        rt = None if ranges is None else ranges.t()
        gx = gy = gv = None
        if ctx.needs_input_grad[1]:
            if ctx.unit is not None:
                gx = (g.unsqueeze(-1) * ctx.unit).reshape(xs).to(xdt)
            else:
                gx = _KernelConv._row_gradient(kind, xb, yb, vb, g, blur, ranges, flags).reshape(xs).to(xdt)
        if ctx.needs_input_grad[2]:
            gy = _KernelConv._row_gradient(kind, yb, xb, g, vb, blur, rt, flags).reshape(ys).to(ydt)
        if ctx.needs_input_grad[3]:
            rows = _gauss_compact_rows(kind, yb, xb.shape[1], rt, flags) if (flags & FLAG_GRAD_FAMILY) else None
            if rows is not None:
                gv = _unsort_rows(rows[0], kernel_conv_fwd_raw(kind, rows[1], xb, g, blur, rt, flags))
            else:
                gv = kernel_conv_fwd_raw(kind, yb, xb, g, blur, rt, flags)
            gv = gv.reshape(vs).to(vdt)
        return None, gx, gy, gv, None, None, None, None

    @staticmethod
    def _row_gradient(kind, rows_pts, cols_pts, v, g, blur, ranges, flags):
        """d/d rows of sum_i g_i sum_j k(rows_i, cols_j) v_j (g per row, v per column), big gaussian launches in compact row order."""
        rows = _gauss_compact_rows(kind, rows_pts, cols_pts.shape[1], ranges, flags)
        if rows is None:
            return kernel_conv_bwd_x_raw(kind, rows_pts, cols_pts, v, g, blur, ranges, flags)
        perm, X = rows
        return _unsort_rows(perm, kernel_conv_bwd_x_raw(kind, X, cols_pts, v, g[:, perm].contiguous(), blur, ranges, flags))


# product + row gradient in one pass when x requires gradients (set_kernel_grad_fusion(False): always two reductions)
_fuse_kernel_grad = True


def set_kernel_grad_fusion(enabled):
    global _fuse_kernel_grad
    _fuse_kernel_grad = bool(enabled)


def kernel_grad_fusion():
    return _fuse_kernel_grad


def kernel_conv(kind, x, y, v, blur=0.05, ranges=None, flags=0):
    """Kernel-matrix x vector product on the GPU; ``kind`` is a name or a GLHIP_* code."""
    kind = KERNEL_KINDS[kind] if isinstance(kind, str) else int(kind)
    return _KernelConv.apply(kind, x, y, v, 1.0 if blur is None else float(blur), ranges, int(flags) | ENV_FLAGS,
                             torch.is_grad_enabled() and x.requires_grad)


def kernel_conv_with_unit(kind, x, y, v, blur, want_unit, flags=0):
    """No autograd: ``(K v, d (K v)_i / d x_i | None)`` from one reduction where a product-and-gradient kernel exists (the building
    block of kernel_samples._UnionNorm); shapes follow x: (N,), (N,D) or (B,N), (B,N,D)."""
    kind = KERNEL_KINDS[kind] if isinstance(kind, str) else int(kind)
    with torch.no_grad():
        _, _, _, batched, out, unit = _KernelConv.product(kind, x, y, v, 1.0 if blur is None else float(blur), None, int(flags) | ENV_FLAGS,
                                                          want_unit)
    if not batched:
        out, unit = out.view(-1), (None if unit is None else unit.view(-1, unit.shape[-1]))
    return out, unit


def kernel_conv_row_gradient(kind, x, y, v, g, blur, flags=0):
    """No autograd: d/dx of sum_i g_i (K v)_i as one gradient reduction (``glhip_kernel_conv_bwd_x``), shaped like x."""
    kind = KERNEL_KINDS[kind] if isinstance(kind, str) else int(kind)
    with torch.no_grad():
        xb, yb, vb, _ = _as_batched(_points(x, "x", True), _points(y, "y", True), v.detach().contiguous())
        if yb.dtype != xb.dtype:
            yb = yb.to(xb.dtype)
        gb = _vec(g, xb).reshape(xb.shape[0], -1)
        return _KernelConv._row_gradient(kind, xb, yb, vb, gb, 1.0 if blur is None else float(blur), None, int(flags) | ENV_FLAGS).reshape(x.shape)


class _SoftminDense(torch.autograd.Function):
    """Row-wise soft-min of an explicit cost matrix (the tensorized backend on GPU tensors)."""

    @staticmethod
    def forward(ctx, C, h, eps):
        Cb, hb = C.float().contiguous(), h.float().contiguous()
        out = softmin_dense_fwd_raw(Cb, hb, eps)
        ctx.save_for_backward(Cb, hb, out)
        ctx.eps = eps
        ctx.dtypes = (C.dtype, h.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        Cb, hb, out = ctx.saved_tensors
        eps = ctx.eps
        # d out_i / d C_ij = P_ij,  d out_i / d h_j = -eps P_ij,  P = softmax_j(h_j - C_ij/eps)
        P = torch.exp(hb[:, None, :] - Cb / eps + (out / eps)[:, :, None]) * grad_out[:, :, None]
        gC = P.to(ctx.dtypes[0]) if ctx.needs_input_grad[0] else None
        gh = (-eps * P.sum(1)).to(ctx.dtypes[1]) if ctx.needs_input_grad[1] else None
        return gC, gh, None


def softmin_dense(eps, C, h):
    return _SoftminDense.apply(C, h, float(eps))


class _LseLines(torch.autograd.Function):
    """out[..., i] = log sum_j exp(h[..., j] - c(i, j)) along the last axis of a contiguous fp32 tensor, c the (scaled)
    squared or absolute distance between grid samples i/N and j/N (``include/glhip.h``); differentiable in h."""

    @staticmethod
    def forward(ctx, h, eps, p):
        if not h.is_cuda:
            raise RuntimeError("geomloss_amd: the grid soft-min runs on the HIP kernels only; move the images to a GPU.")
        hc = h.detach().float().contiguous()
        N = hc.shape[-1]
        R = hc.numel() // max(N, 1)
        out = torch.empty_like(hc)
        lib = load_library()
        with torch.cuda.device(hc.device):
            rc = lib.glhip_lse_lines_fwd(hc.data_ptr(), out.data_ptr(), R, N, float(eps), int(p), _stream(hc))
        _check(rc, lib)
        ctx.save_for_backward(hc, out)
        ctx.cfg = (float(eps), int(p), h.dtype)
        return out.to(h.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        hc, out = ctx.saved_tensors
        eps, p, dtype = ctx.cfg
        g = grad_out.float().contiguous()
        gh = torch.empty_like(hc)
        N = hc.shape[-1]
        lib = load_library()
        with torch.cuda.device(hc.device):
            rc = lib.glhip_lse_lines_bwd(hc.data_ptr(), out.data_ptr(), g.data_ptr(), gh.data_ptr(), hc.numel() // max(N, 1), N,
                                         eps, p, _stream(hc))
        _check(rc, lib)
        return gh.to(dtype), None, None


def cmin(x, y, g, p=2, ranges=None, flags=0):
    """Hard C-transform on the GPU: min_j [C(x_i, y_j) - g_j], C = |x-y|^2/2 (p = 2) or |x-y| (p = 1); not differentiable.
    x: (N,D)|(B,N,D), y: (M,D)|(B,M,D), g: (M,)|(B,M) -> (N,)|(B,N) fp32."""
    lib = load_library()
    xb, yb, gb, batched = _as_batched(_points(x.detach(), "x"), _points(y.detach(), "y"), _f32(g))
    if yb.dtype != xb.dtype:
        yb = yb.to(xb.dtype)
    B, N, D = xb.shape
    M = yb.shape[1]
    out = torch.empty((B, N), dtype=torch.float32, device=xb.device)
    with torch.cuda.device(xb.device):
        ws, ws_args = _workspace(lib, xb, B, N, M, D, ranges)
        rc = lib.glhip_cmin_fwd(xb.data_ptr(), yb.data_ptr(), gb.data_ptr(), out.data_ptr(), B, N, M, D, int(p), _dtype_code(xb),
                                *_range_args(ranges, B), *ws_args, int(flags) | ENV_FLAGS, _stream(xb))
    _check(rc, lib)
    return out if batched else out.view(-1)


def max_lines(g, step, p=2):
    """out[..., i] = max_j [g[..., j] - c(i, j)] along the last axis, c = (step (i-j))^2 (p = 2) or step |i-j| (p = 1)."""
    if not g.is_cuda:
        raise RuntimeError("geomloss_amd: the grid C-transform runs on the HIP kernels only; move the arrays to a GPU.")
    gc = g.detach().float().contiguous()
    N = gc.shape[-1]
    out = torch.empty_like(gc)
    lib = load_library()
    with torch.cuda.device(gc.device):
        rc = lib.glhip_max_lines_fwd(gc.data_ptr(), out.data_ptr(), gc.numel() // max(N, 1), N, float(step), int(p), _stream(gc))
    _check(rc, lib)
    return out.to(g.dtype)


def lse_lines(h, eps, p=2):
    return _LseLines.apply(h, float(eps), int(p))


def settle_host():
    """For latency-critical loops (serving, benchmarks): one full Python garbage collection now, and its survivors moved out of the
    collector's reach (``gc.freeze()``).  The first generation-2 collection of a process that has imported torch walks ~1e6 objects
    and takes 30-40 ms; it fires once, a few thousand container allocations in, wherever the program happens to be — and while the
    host is stopped the launch queue of a 2 ms loss drains and the GPU idles (measured: profiles/r05_shard_stall.txt; this was the
    intermittent 2x outlier of the B = 32 shard of BASELINE configs[3]).  Call it after the warm-up of such a loop; bench.py does."""
    import gc
    gc.collect()
    gc.freeze()


# ----------------------------------------------------------------------------------------------
#  hipGraph capture of launch-bound loops
# ----------------------------------------------------------------------------------------------

class GraphCache:
    """Captures ``fn(*tensors) -> tuple of tensors`` into a hipGraph (``torch.cuda.CUDAGraph``) the first time a key is
    seen and replays it afterwards.  Everything ``fn`` launches — our C-ABI kernels included: they are issued on torch's
    current stream, which is the capture stream — becomes one graph launch.  Inputs are copied into static buffers,
    outputs are cloned out of them.  Small LRU: a graph pins its memory pool."""

    def __init__(self, capacity=8):
        self.capacity, self.entries = capacity, {}

    def run(self, key, fn, inputs):
        entry = self.entries.pop(key, None)
        if entry is None:
            static_in = [t.detach().clone() for t in inputs]
            side = torch.cuda.Stream(device=static_in[0].device)
            side.wait_stream(torch.cuda.current_stream(static_in[0].device))
            with torch.cuda.stream(side):      # warm-up outside capture (loads code objects, sizes the allocator pools)
                fn(*static_in)
            torch.cuda.current_stream(static_in[0].device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = fn(*static_in)
            entry = (graph, static_in, static_out)
            if len(self.entries) >= self.capacity:
                self.entries.pop(next(iter(self.entries)))
        else:
            for dst, src in zip(entry[1], inputs):
                dst.copy_(src)
        self.entries[key] = entry   # most recently used last
        entry[0].replay()
        return tuple(None if o is None else o.clone() for o in entry[2])
