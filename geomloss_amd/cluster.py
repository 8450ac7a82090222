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
    """
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
