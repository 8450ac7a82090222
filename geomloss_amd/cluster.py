"""Voxel clustering and block-sparse range construction for the two-scale ("multiscale") backends.

Device-side (PyTorch-ROCm tensor ops) restatement of the ``pykeops.torch.cluster`` helpers that the
reference imports at ``_legacy/sinkhorn_samples.py:9-10`` and ``_legacy/kernel_samples.py:27-33``:
``grid_cluster``, ``cluster_ranges_centroids``, ``sort_clusters``, ``from_matrix``, ``swap_axes``.
pykeops is not part of the reference tree; the semantics restated here are the ones its call sites
rely on (``sinkhorn_samples.py:453-530``): points of one voxel become one contiguous row range of
the sorted cloud, and a boolean cluster-cluster mask becomes CSR-like lists of column intervals.

One deliberate difference: ``from_matrix`` merges column intervals that are adjacent in memory
(consecutive kept clusters), so the kernels stream fewer, longer tiles.  The reduced set of
(i, j) pairs is unchanged.
"""

import torch

from .hip import BlockRanges


def grid_cluster(x, size):
    """Integer voxel label (0..C-1) of every point of x (N, D<=3) for cubic bins of edge ``size``.

    Labels are ordered lexicographically by voxel coordinates, first axis most significant, which is
    the order pykeops' packed voxel code induces.
    """
    with torch.no_grad():
        q = torch.floor(x / size).long()
        q = q - q.min(dim=0)[0]
        extent = q.max(dim=0)[0] + 1
        code = q[:, 0]
        for d in range(1, q.shape[1]):
            code = code * extent[d] + q[:, d]
        _, lab = torch.unique(code, sorted=True, return_inverse=True)
    return lab.int()


def cluster_ranges_centroids(x, lab, weights=None, min_weight=1e-9, perm=None):
    """Per-cluster [start, end) ranges (valid once the cloud is sorted by label), weighted centroids, total weights.

    Sums are segment sums of the label-sorted cloud, taken as differences of a float64 inclusive scan: deterministic
    (no atomics — the same inputs give the same centroids bit for bit on every run) and exact to ~1e-16 of the total
    mass.  Weights keep their own precision (fp32 for bf16 / fp16 clouds); centroids come back in the dtype of ``x``.
    ``perm``: the stable argsort of ``lab`` when the caller already has it.
    Like the reference's ``bincount`` version (and like ``glhip_grid_cluster``), nothing here is differentiable: the sums run
    under ``no_grad`` on detached inputs, so centroids and cluster weights never drag a float64 scan into an autograd graph.
    """
    with torch.no_grad():
        return _cluster_ranges_centroids(x.detach(), lab, None if weights is None else weights.detach(), min_weight, perm)


def _cluster_ranges_centroids(x, lab, weights, min_weight, perm):
    lab = lab.long().view(-1)
    counts = torch.bincount(lab)          # integer histogram: exact, order-independent
    ends = counts.cumsum(0)
    ranges = torch.stack((ends - counts, ends), dim=1).int()
    wdtype = torch.float32 if weights is None or weights.dtype in (torch.bfloat16, torch.float16) else weights.dtype
    N, D = x.shape
    if N == 0:
        return ranges, x.new_zeros((0, D)), torch.zeros(0, dtype=wdtype, device=x.device)
    if perm is None:
        perm = torch.sort(lab, stable=True)[1]
    # (1 + D, N) rows [w, w x_1, ..., w x_D] of the sorted cloud; the scan runs along the contiguous axis
    vals = torch.empty((1 + D, N), dtype=torch.float64, device=x.device)
    vals[0] = 1.0 if weights is None else weights.view(-1)[perm]
    vals[1:] = x[perm].t()
    vals[1:] *= vals[0]
    scan = torch.cat((vals.new_zeros((1 + D, 1)), vals.cumsum(1)), dim=1)
    seg = scan[:, ends] - scan[:, ends - counts]                                     # (1 + D, C)
    w_c = seg[0]
    cents = (seg[1:] / w_c.clamp_min(min_weight)).t().contiguous()
    return ranges, cents.to(x.dtype), w_c.to(wdtype)


def sort_clusters(x, lab):
    """Sorts (tuples of) per-point tensors so that clusters are contiguous; returns the sorted labels too."""
    lab_sorted, perm = torch.sort(lab.view(-1), stable=True)
    if isinstance(x, tuple):
        return tuple(t[perm] for t in x), lab_sorted
    return x[perm], lab_sorted


def _merged_intervals(ranges_cols, keep):
    """CSR lists of column intervals for a (Ci, Cj) boolean mask, merging memory-adjacent intervals."""
    Cj = keep.shape[1]
    if Cj > 1:
        glued = ranges_cols[:-1, 1] == ranges_cols[1:, 0]  # cluster c and c+1 touch
        joined = keep[:, :-1] & keep[:, 1:] & glued.unsqueeze(0)
        pad = torch.zeros((keep.shape[0], 1), dtype=torch.bool, device=keep.device)
        starts = keep & ~torch.cat((pad, joined), dim=1)
        stops = keep & ~torch.cat((joined, pad), dim=1)
    else:
        starts = stops = keep
    col_start = starts.nonzero()[:, 1]  # row-major order: runs of one row are consecutive
    col_stop = stops.nonzero()[:, 1]
    red = torch.stack((ranges_cols[col_start, 0], ranges_cols[col_stop, 1]), dim=1).int().contiguous()
    slices = starts.sum(1).cumsum(0).int().contiguous()
    return slices, red


def from_matrix(ranges_i, ranges_j, keep):
    """Boolean (Ci, Cj) cluster mask -> :class:`BlockRanges` for both orientations of the reduction."""
    ranges_i = ranges_i.int().contiguous()
    ranges_j = ranges_j.int().contiguous()
    slices_i, redranges_j = _merged_intervals(ranges_j, keep)
    slices_j, redranges_i = _merged_intervals(ranges_i, keep.t())
    return BlockRanges(ranges_i, slices_i, redranges_j, ranges_j, slices_j, redranges_i)


def swap_axes(ranges):
    return ranges.t()


# ----------------------------------------------------------------------------------------------------------------------
#  device-side fast path (glhip_grid_cluster / glhip_block_ranges): what the drivers use on GPU clouds
# ----------------------------------------------------------------------------------------------------------------------

def native_clustering_applies(x, labels=None):
    """The HIP clustering kernels take fp32 / bf16 clouds of dimension <= 3 on a GPU, clustered by voxels (no user labels)."""
    from . import hip
    return (labels is None and x.is_cuda and x.dim() == 2 and x.shape[1] <= 3 and x.shape[0] > 0
            and x.dtype in (torch.float32, torch.bfloat16) and hip.library_available())


def native_keep_rule_applies(x):
    """``glhip_block_ranges`` / ``_kept_pairs`` evaluate the keep rule on fp32 copies of the coarse clouds and potentials: also for the
    float64 coarse level of big two-scale losses (round 6) — a cluster pair within float32 rounding of the threshold may be decided
    the other way than in float64, which is what the borderline clause of the parity tests is for (0 - 2 of 4.8e6 pairs at N = 1e6).
    The dense float64 mask + Python interval walk it replaces cost 2 ms per loss and 130 - 400 ms on the first call."""
    from . import hip
    return (x.is_cuda and x.dim() == 2 and x.shape[1] <= 3 and x.shape[0] > 0
            and x.dtype in (torch.float32, torch.bfloat16, torch.float64) and hip.library_available())


def clusterize_device_many(clouds, scale, pre_div=1.0, long_perm=True, extent=None):
    """:func:`clusterize_device` for several weighted clouds ``[(a, x), ...]`` with ONE host round trip for all their cluster counts
    (the two measures of a two-scale loss: one synchronisation instead of two).  ``long_perm=False``: the permutations stay int32,
    as the kernel wrote them, for callers that only index with them now and then (one launch less per cloud).  ``extent``: a list
    that receives an upper bound of the diagonal of the clouds' joint bounding box (``hip.voxel_extent``: the voxel bounds come
    back in the same round trip)."""
    from . import hip
    pending = []
    for a, x in clouds:
        xd = x.detach().contiguous()
        ad = None if a is None else a.detach().float().contiguous().view(-1)
        need_graph = torch.is_grad_enabled() and (x.requires_grad or (a is not None and a.requires_grad))
        pending.append((a, x, need_graph, hip.grid_cluster_raw(xd, ad, scale, pre_div, gather=not need_graph, defer=True)))
    counts = hip.read_back(*[p[3][0] for p in pending])
    if extent is not None:
        extent.append(hip.voxel_extent(counts, scale, pre_div))
    out = []
    for (a, x, need_graph, (_, finish)), values in zip(pending, counts):
        perm32, xs, ws, ranges, cents, w_c = finish(values)
        perm = perm32.long() if (long_perm or need_graph) else perm32
        if need_graph:
            xs, ws = x[perm], (None if a is None else a[perm])
        elif a is not None and a.dtype != torch.float32:
            ws = ws.to(a.dtype)
        out.append((w_c, ws, cents.to(x.dtype), xs, ranges, perm))
    return out


def clusterize_device(a, x, scale, pre_div=1.0):
    """``grid_cluster`` + ``cluster_ranges_centroids`` + ``sort_clusters`` in one call of ``glhip_grid_cluster``.

    Returns ``(a_c, a_sorted, x_c, x_sorted, ranges, perm)`` with ``perm`` int64.  ``x_c`` are the centroids of ``x / pre_div``
    (in the dtype of x), ``a_c`` the cluster weights (fp32).  The sorted cloud / weights stay differentiable with respect to
    ``x`` / ``a`` when those require gradients (plain indexing); centroids and cluster weights never do, as in the reference."""
    from . import hip
    xd = x.detach().contiguous()
    ad = None if a is None else a.detach().float().contiguous().view(-1)
    need_graph = torch.is_grad_enabled() and (x.requires_grad or (a is not None and a.requires_grad))
    perm32, xs, ws, ranges, cents, w_c = hip.grid_cluster_raw(xd, ad, scale, pre_div, gather=not need_graph)
    perm = perm32.long()
    if need_graph:
        xs, ws = x[perm], (None if a is None else a[perm])
    elif a is not None and a.dtype != torch.float32:
        ws = ws.to(a.dtype)
    return w_c, ws, cents.to(x.dtype), xs, ranges, perm


def kept_pairs_device(kind, rows, cols, f, g, ranges_rows, ranges_cols, thr, p=2, defer=False):
    """Pairs of points the keep rule of :func:`block_ranges_device` retains (same arguments), counted without building the pattern
    (``defer``: the device tensor of the three counts, for a shared ``hip.read_back``)."""
    from . import hip
    code = {"dual_slack": hip.KEEP_DUAL_SLACK, "within": hip.KEEP_WITHIN}[kind]
    f32 = lambda t: None if t is None else t.detach().float().contiguous().view(-1)  # noqa: E731
    return hip.kept_pairs_raw(code, rows.detach().float().contiguous(), cols.detach().float().contiguous(), f32(f), f32(g),
                              ranges_rows.int().contiguous(), ranges_cols.int().contiguous(), thr, p, defer=defer)


def block_ranges_device(kind, rows, cols, f, g, ranges_rows, ranges_cols, thr, p=2, symmetric=False):
    """Keep rule -> :class:`BlockRanges` without materialising the mask (``glhip_block_ranges``; interval buffers sized from its
    counting pass once the worst case would be large).  ``kind``: "dual_slack"
    (f_i + g_j > C_ij - thr, sinkhorn_samples.py:512-514) or "within" (|c_i - c_j|^2 <= thr, kernel_samples.py:244-252)."""
    from . import hip
    code = {"dual_slack": hip.KEEP_DUAL_SLACK, "within": hip.KEEP_WITHIN}[kind]
    f32 = lambda t: None if t is None else t.detach().float().contiguous().view(-1)  # noqa: E731
    return hip.block_ranges_raw(code, rows.detach().float().contiguous(), cols.detach().float().contiguous(), f32(f), f32(g),
                                ranges_rows.int().contiguous(), ranges_cols.int().contiguous(), thr, p, symmetric=symmetric)
