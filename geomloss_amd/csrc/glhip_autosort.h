// glhip_autosort.h — big dense DISTANCE reductions sort their clouds themselves (round 6; round-5 review, missing #4).
//
// The matrix-core distance kernel of glhip_dist_x32.h (p = 1 soft-min, laplacian / energy products, D <= 3) needs spatially compact
// row blocks.  Until round 5 only the Python side knew how to provide them (voxel sort + boustrophedon chaining, hip.py:_CompactRows),
// so a caller binding glhip_softmin_fwd(p = 1) through INTEGRATION.md's stub — the `lse_genred("Norm2(X-Y)")` call site,
// _legacy/sinkhorn_samples.py:316-334 — got the generic explicit-difference kernel: 346 ms instead of 191 at N = M = 1e6.
// Now the entry points do it behind the ABI, inside the caller's workspace and on the caller's stream, without a host round trip:
//   bounding box -> voxel edge (~256 rows per voxel; columns: 512) -> boustrophedon path index of every point as the sort key ->
//   rocPRIM radix sort -> gathered clouds and column / row vectors -> the block-sparse launch "every slab of 256 rows x all columns"
//   with GLHIP_FLAG_MFMA_DIST -> results scattered back to the caller's row order.
// Conditions: B = 1, dense, D <= 3, N >= 65536, N M >= 5e8, neither GLHIP_FLAG_NO_MFMA / _DIRECT nor GLHIP_FLAG_NO_SORT, and a
// workspace of glhip_workspace_bytes(...) (smaller: the generic kernel, as before).  Two sorts of ~0.5 ms against ~200 ms.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace glhip {

constexpr int kSortSlab = 256;        // rows per row block = the row tile of the distance kernel (8 wavefronts x 32 rows)
constexpr int kSortColChunks = 8;     // column intervals per slab: something for the column splits to split
constexpr int kSortRowsPerVoxel = 256;

inline size_t as_align256(size_t b) { return (b + 255) & ~(size_t)255; }

// implemented in glhip_cluster.hip (rocPRIM lives there)
size_t compact_sort_scratch_bytes(int n);
int compact_sort(const void* z, int n, int D, int in_dtype, int rows_per_voxel, int32_t* perm, void* z_sorted, void* scratch,
                 size_t scratch_bytes, hipStream_t st);
void gather_f32(const float* src, const int32_t* perm, float* dst, int n, hipStream_t st);       // dst[k] = src[perm[k]]
void scatter_f32(const float* src, const int32_t* perm, float* dst, int n, hipStream_t st, int width = 1);      // dst[perm[k], :] = src[k, :]
void slab_ranges(int N, int M, int32_t* ranges_i, int32_t* slices_i, int32_t* red, hipStream_t st);

inline bool autosort_applies(int B, int N, int M, int D, int n_ranges, int flags) {
    return B == 1 && n_ranges == 0 && D <= 3 && N >= 65536 && (double)N * M >= 5e8 &&
           !(flags & (2 /* NO_MFMA */ | 1 /* DIRECT */ | 512 /* NO_SORT */));
}

struct AutoSort {
    bool on = false;
    int C = 0;                              // slabs
    int32_t *perm_x = nullptr, *perm_y = nullptr, *ranges_i = nullptr, *slices_i = nullptr, *red = nullptr;
    void *xs = nullptr, *ys = nullptr;
    float *col0 = nullptr, *col1 = nullptr, *row0 = nullptr, *out = nullptr;      // gathered per-column / per-row vectors, sorted output
    float* out_rows = nullptr;              // (N, D) sorted output (row gradients)
    void* inner_ws = nullptr;
    size_t inner_bytes = 0;
};

// bytes the sorted call carves off the FRONT of the workspace (everything but the inner launch's own scratch)
inline size_t autosort_bytes(int N, int M, int D) {
    const int C = (N + kSortSlab - 1) / kSortSlab;
    const int L = N > M ? N : M;
    return as_align256((size_t)N * 4) + as_align256((size_t)M * 4) + as_align256((size_t)N * D * 4) + as_align256((size_t)M * D * 4) +
           2 * as_align256((size_t)M * 4) + 2 * as_align256((size_t)N * 4) + as_align256((size_t)N * D * 4) + as_align256((size_t)C * 8) +
           as_align256((size_t)C * 4) +
           as_align256((size_t)C * kSortColChunks * 8) + compact_sort_scratch_bytes(L);
}

// Sorts both clouds into the workspace; `a.on` stays false when the workspace is too small for the sorted call plus `inner_min`
// bytes of scratch for the launch itself (the caller then runs the generic kernel).
inline int autosort_prepare(AutoSort& a, const void* x, const void* y, int N, int M, int D, int in_dtype, void* workspace,
                            size_t workspace_bytes, size_t inner_min, hipStream_t st) {
    const size_t need = autosort_bytes(N, M, D);
    if (!workspace || workspace_bytes < need + inner_min) return 0;
    char* w = static_cast<char*>(workspace);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = w + off; off += as_align256(bytes); return p; };
    a.C = (N + kSortSlab - 1) / kSortSlab;
    a.perm_x = reinterpret_cast<int32_t*>(take((size_t)N * 4));
    a.perm_y = reinterpret_cast<int32_t*>(take((size_t)M * 4));
    a.xs = take((size_t)N * D * 4);
    a.ys = take((size_t)M * D * 4);
    a.col0 = reinterpret_cast<float*>(take((size_t)M * 4));
    a.col1 = reinterpret_cast<float*>(take((size_t)M * 4));
    a.row0 = reinterpret_cast<float*>(take((size_t)N * 4));
    a.out = reinterpret_cast<float*>(take((size_t)N * 4));
    a.out_rows = reinterpret_cast<float*>(take((size_t)N * D * 4));
    a.ranges_i = reinterpret_cast<int32_t*>(take((size_t)a.C * 8));
    a.slices_i = reinterpret_cast<int32_t*>(take((size_t)a.C * 4));
    a.red = reinterpret_cast<int32_t*>(take((size_t)a.C * kSortColChunks * 8));
    const int L = N > M ? N : M;
    const size_t sb = compact_sort_scratch_bytes(L);
    void* scratch = take(sb);
    int rc = compact_sort(x, N, D, in_dtype, kSortRowsPerVoxel, a.perm_x, a.xs, scratch, sb, st);
    if (rc) return rc;
    rc = compact_sort(y, M, D, in_dtype, 2 * kSortRowsPerVoxel, a.perm_y, a.ys, scratch, sb, st);
    if (rc) return rc;
    slab_ranges(N, M, a.ranges_i, a.slices_i, a.red, st);
    a.inner_ws = w + off;
    a.inner_bytes = workspace_bytes - off;
    a.on = true;
    return 0;
}

}  // namespace glhip
