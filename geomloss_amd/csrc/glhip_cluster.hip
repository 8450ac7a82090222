// glhip_cluster.hip — the cluster pyramid of the two-scale ("multiscale") backends, on the device:
//   glhip_grid_cluster   voxel labels -> cluster-sorted cloud, per-cluster row ranges, weighted centroids, cluster weights
//   glhip_block_ranges   cluster-cluster keep rule -> CSR lists of merged column intervals, both orientations
// The reference does this with pykeops.torch.cluster.{grid_cluster, cluster_ranges_centroids, sort_clusters, from_matrix}
// (torch ops; _legacy/sinkhorn_samples.py:453-530, _legacy/kernel_samples.py:214-256); geomloss_amd/cluster.py restates those
// helpers with torch tensor ops (dozens of small launches, two host round trips each, and 1.2 s of lazily loaded torch code
// objects on the first call).  Here: a handful of launches on the caller's stream, no host round trip inside the library,
// deterministic results (fixed-order float64 sums, no atomics on floating-point data).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <climits>
#include <cstdint>

#include "glhip_autosort.h"
#include "glhip_common.h"
#include "glhip_error.h"

using namespace glhip;

namespace {

constexpr int kAxisBits = 21;                       // voxel coordinates (relative to the cloud's minimum) per axis
constexpr int kAxisMax = (1 << kAxisBits) - 1;

struct ClusterHead {        // first bytes of the workspace
    int qmin[3];
    int qmax[3];
    int overflow;           // a voxel coordinate did not fit kAxisBits
    int pad;
};

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

template <typename T>
__device__ __forceinline__ int voxel_of(const T* __restrict__ x, long idx, float pre_div, float voxel) {
    // two IEEE divisions, as torch evaluates floor((x / blur) / size)
    return (int)floorf((to_f32<T>(x[idx]) / pre_div) / voxel);
}

// the workspace head in its start state (one launch; three hipMemsetAsync calls were four fill kernels)
__global__ void head_init_kernel(ClusterHead* head) {
    const int t = threadIdx.x;
    if (t < 3) head->qmin[t] = INT_MAX;
    else if (t < 6) head->qmax[t - 3] = INT_MIN;
    else if (t == 6) head->overflow = 0;
    else if (t == 7) head->pad = 0;
}

template <typename T>
__global__ void __launch_bounds__(256) bounds_kernel(const T* __restrict__ x, int N, int D, float pre_div, float voxel, ClusterHead* head) {
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
        for (int d = 0; d < D; ++d) {
            const int q = voxel_of<T>(x, i * D + d, pre_div, voxel);
            lo[d] = min(lo[d], q);
            hi[d] = max(hi[d], q);
        }
    }
    for (int d = 0; d < D; ++d) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = min(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = max(hi[d], __shfl_xor(hi[d], off, 64));
        }
    }
    // one pair of atomics per axis and WORKGROUP: 4096 wavefronts hammering the same 6 words took 0.29 ms at 1e6 points
    __shared__ int wlo[4][3], whi[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int d = 0; d < D; ++d) { wlo[wave][d] = lo[d]; whi[wave][d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < D) {
        const int d = threadIdx.x;
        int l = wlo[0][d], h = whi[0][d];
        for (int w = 1; w < 4; ++w) { l = min(l, wlo[w][d]); h = max(h, whi[w][d]); }
        if (l <= h) {
            atomicMin(&head->qmin[d], l);
            atomicMax(&head->qmax[d], h);
        }
    }
}

// key = voxel coordinates packed most-significant-axis first: sorting the keys sorts the voxels lexicographically, which is
// the order of the labels pykeops' grid_cluster hands out
template <typename T>
__global__ void __launch_bounds__(256) keys_kernel(const T* __restrict__ x, int N, int D, float pre_div, float voxel, ClusterHead* head,
                                                   uint64_t* __restrict__ keys, int32_t* __restrict__ idx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    uint64_t key = 0;
    for (int d = 0; d < D; ++d) {
        const long q = (long)voxel_of<T>(x, i * D + d, pre_div, voxel) - head->qmin[d];
        if (q > kAxisMax) head->overflow = 1;
        key = (key << kAxisBits) | (uint64_t)(q & kAxisMax);
    }
    keys[i] = key;
    idx[i] = (int32_t)i;
}

__global__ void __launch_bounds__(256) flags_kernel(const uint64_t* __restrict__ keys, int N, int32_t* __restrict__ flags) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < N) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// labels (inclusive scan of the flags) -> [start, end) of every cluster, and the cluster count
__global__ void __launch_bounds__(256) ranges_kernel(const int32_t* __restrict__ incl, int N, int32_t* __restrict__ ranges, int32_t* n_clusters,
                                                     const ClusterHead* __restrict__ head) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int c = incl[i] - 1;
    if (i == 0 || incl[i - 1] != incl[i]) {
        ranges[2 * c] = (int32_t)i;
        if (c > 0) ranges[2 * (c - 1) + 1] = (int32_t)i;
    }
    if (i == N - 1) {
        ranges[2 * c + 1] = N;
        n_clusters[0] = c + 1;
        n_clusters[1] = head->overflow;     // the overflow flag of keys_kernel travels with the cluster count
        for (int d = 0; d < 3; ++d) {       // ... and so do the voxel bounds of the cloud: its bounding box, to one voxel
            n_clusters[2 + d] = head->qmin[d] == INT_MAX ? 0 : head->qmin[d];
            n_clusters[5 + d] = head->qmax[d] == INT_MIN ? 0 : head->qmax[d];
        }
    }
}

__device__ __forceinline__ double shfl_xor_f64(double v, int off) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, off, 64);
    hi = __shfl_xor(hi, off, 64);
    return __hiloint2double(hi, lo);
}

// one wavefront per cluster: weight and weighted centroid of (x / pre_div) in float64, in a fixed order
template <typename T>
__global__ void __launch_bounds__(256) centroids_kernel(const T* __restrict__ x, const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ ranges, const int32_t* __restrict__ n_clusters, int D,
                                                        float pre_div, float* __restrict__ centroids, float* __restrict__ weights_c) {
    const int lane = threadIdx.x & 63;
    const int n_waves = gridDim.x * 4;
    const int C = n_clusters[0];
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < C; c += n_waves) {
        const int r0 = ranges[2 * c], r1 = ranges[2 * c + 1];
        double sw = 0.0, sx[3] = {0.0, 0.0, 0.0};
        for (int p = r0 + lane; p < r1; p += 64) {
            const long j = perm[p];
            const double wj = w ? (double)w[j] : 1.0;
            sw += wj;
            for (int d = 0; d < D; ++d) sx[d] += wj * (double)(to_f32<T>(x[j * D + d]) / pre_div);
        }
        for (int off = 32; off > 0; off >>= 1) {
            sw += shfl_xor_f64(sw, off);
            for (int d = 0; d < D; ++d) sx[d] += shfl_xor_f64(sx[d], off);
        }
        if (lane == 0) {
            const double den = sw > 1e-9 ? sw : 1e-9;     // cluster_ranges_centroids' min_weight
            for (int d = 0; d < D; ++d) centroids[(long)c * D + d] = (float)(sx[d] / den);
            weights_c[c] = (float)sw;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gather_kernel(const T* __restrict__ x, const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                     int N, int D, T* __restrict__ x_sorted, float* __restrict__ w_sorted) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    const long j = perm[p];
    for (int d = 0; d < D; ++d) x_sorted[p * D + d] = x[j * D + d];
    if (w_sorted) w_sorted[p] = w ? w[j] : 1.0f;
}

size_t sort_temp_bytes(int N) {
    size_t bytes = 0;
    uint64_t* k = nullptr;
    int32_t* v = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)N, 0u, (unsigned)(3 * kAxisBits), (hipStream_t)0) != hipSuccess || bytes == 0)
        bytes = (size_t)N * 16 + (1 << 20);   // no device to ask (build box): a bound rocPRIM's merge/onesweep paths stay under
    (void)hipGetLastError();
    return bytes;
}

size_t scan_temp_bytes(int N) {
    size_t bytes = 0;
    int32_t* v = nullptr;
    if (rocprim::inclusive_scan(nullptr, bytes, v, v, (size_t)N, rocprim::plus<int32_t>(), (hipStream_t)0) != hipSuccess || bytes == 0)
        bytes = (size_t)N * 4 + (1 << 20);
    (void)hipGetLastError();
    return bytes;
}

template <typename T>
int grid_cluster_typed(const void* x_, const float* w, int N, int D, float pre_div, float voxel, int32_t* perm, void* x_sorted,
                       float* w_sorted, int32_t* ranges, float* centroids, float* weights_c, int32_t* n_clusters, void* workspace,
                       size_t workspace_bytes, hipStream_t st) {
    const T* x = static_cast<const T*>(x_);
    char* ws = static_cast<char*>(workspace);
    ClusterHead* head = reinterpret_cast<ClusterHead*>(ws);
    size_t off = align256(sizeof(ClusterHead));
    uint64_t* keys_in = reinterpret_cast<uint64_t*>(ws + off); off += align256((size_t)N * 8);
    uint64_t* keys_out = reinterpret_cast<uint64_t*>(ws + off); off += align256((size_t)N * 8);
    int32_t* idx_in = reinterpret_cast<int32_t*>(ws + off); off += align256((size_t)N * 4);
    int32_t* flags = reinterpret_cast<int32_t*>(ws + off); off += align256((size_t)N * 4);
    size_t temp_sort = sort_temp_bytes(N), temp_scan = scan_temp_bytes(N);
    const size_t temp = temp_sort > temp_scan ? temp_sort : temp_scan;
    if (off + temp > workspace_bytes)
        return fail(GLHIP_EINVAL, "glhip_grid_cluster: workspace of %zu bytes, need %zu (glhip_cluster_workspace_bytes)", workspace_bytes, off + temp);
    void* tmp = ws + off;

    const int blocks = (N + 255) / 256;
    hipLaunchKernelGGL(head_init_kernel, dim3(1), dim3(64), 0, st, head);
    hipLaunchKernelGGL((bounds_kernel<T>), dim3(blocks < 512 ? blocks : 512), dim3(256), 0, st, x, N, D, pre_div, voxel, head);
    hipLaunchKernelGGL((keys_kernel<T>), dim3(blocks), dim3(256), 0, st, x, N, D, pre_div, voxel, head, keys_in, idx_in);
    // rocPRIM's radix sort is stable: equal keys (one voxel) keep the order of their indices, as torch.sort(stable=True) does
    if (rocprim::radix_sort_pairs(tmp, temp_sort, keys_in, keys_out, idx_in, perm, (size_t)N, 0u, (unsigned)(D * kAxisBits), st) != hipSuccess)
        return fail(GLHIP_ELAUNCH, "glhip_grid_cluster: radix sort failed: %s", hipGetErrorString(hipGetLastError()));
    hipLaunchKernelGGL(flags_kernel, dim3(blocks), dim3(256), 0, st, keys_out, N, flags);
    int32_t* incl = idx_in;   // the unsorted indices are no longer needed
    if (rocprim::inclusive_scan(tmp, temp_scan, flags, incl, (size_t)N, rocprim::plus<int32_t>(), st) != hipSuccess)
        return fail(GLHIP_ELAUNCH, "glhip_grid_cluster: scan failed: %s", hipGetErrorString(hipGetLastError()));
    hipLaunchKernelGGL(ranges_kernel, dim3(blocks), dim3(256), 0, st, incl, N, ranges, n_clusters, head);
    // one wavefront per cluster, the count known on the device only: a wavefront per 4 points covers the fine grids of small clouds
    // in one round (a cluster loop of 7 rounds cost 50 us at N = 1e4), surplus wavefronts leave at once
    const int blocks_c = (N + 15) / 16;
    hipLaunchKernelGGL((centroids_kernel<T>), dim3(blocks_c < 2048 ? blocks_c : 2048), dim3(256), 0, st, x, w, perm, ranges, n_clusters, D, pre_div,
                       centroids, weights_c);
    if (x_sorted)
        hipLaunchKernelGGL((gather_kernel<T>), dim3(blocks), dim3(256), 0, st, x, w, perm, N, D, static_cast<T*>(x_sorted), w_sorted);
    return check_launch("glhip_grid_cluster");
}

// ---- compact order of a cloud (glhip_autosort.h): bounding box -> voxel edge -> boustrophedon path keys -> radix sort -> gather ----

struct SortHead {           // first bytes of the sort scratch
    unsigned lo[3], hi[3];  // bounding box, as order-preserving unsigned images of the floats
    float voxel;
    int qmin[3], ext[3];
};

__device__ __forceinline__ unsigned ordered_of(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_of(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__global__ void sort_head_init_kernel(SortHead* head) {
    const int t = threadIdx.x;
    if (t < 3) { head->lo[t] = 0xFFFFFFFFu; head->hi[t] = 0u; }
}

template <typename T>
__global__ void __launch_bounds__(256) bbox_kernel(const T* __restrict__ x, int n, int D, SortHead* head) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        for (int d = 0; d < D; ++d) {
            const float v = to_f32<T>(x[i * D + d]);
            lo[d] = fminf(lo[d], v);
            hi[d] = fmaxf(hi[d], v);
        }
    }
    for (int d = 0; d < D; ++d) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
        }
    }
    __shared__ float wlo[4][3], whi[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int d = 0; d < D; ++d) { wlo[wave][d] = lo[d]; whi[wave][d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < D) {
        const int d = threadIdx.x;
        float l = wlo[0][d], h = whi[0][d];
        for (int w = 1; w < 4; ++w) { l = fminf(l, wlo[w][d]); h = fmaxf(h, whi[w][d]); }
        if (l <= h) {
            atomicMin(&head->lo[d], ordered_of(l));
            atomicMax(&head->hi[d], ordered_of(h));
        }
    }
}

// voxel edge that puts ~rows_per_voxel points in an occupied voxel (hip.py:_voxel_for), and the voxel grid of the cloud
__global__ void voxel_kernel(SortHead* head, int n, int D, int rows_per_voxel) {
    if (threadIdx.x != 0) return;
    float lo[3], ext[3], emax = 0.f;
    for (int d = 0; d < D; ++d) {
        lo[d] = float_of(head->lo[d]);
        ext[d] = float_of(head->hi[d]) - lo[d];
        emax = fmaxf(emax, ext[d]);
    }
    float vol = 1.f;
    int live = 0;
    for (int d = 0; d < D; ++d)
        if (ext[d] > 1e-6f * fmaxf(emax, 1e-30f)) { vol *= ext[d]; ++live; }
    float voxel = powf(vol * (float)rows_per_voxel / (float)n, 1.0f / (float)(live > 0 ? live : 1));
    voxel = fmaxf(fmaxf(voxel, emax / (float)(1 << 20)), 1e-30f);       // never more than 2^20 voxels along an axis
    head->voxel = voxel;
    for (int d = 0; d < 3; ++d) {
        head->qmin[d] = d < D ? (int)floorf(lo[d] / voxel) : 0;
        head->ext[d] = d < D ? (int)floorf((lo[d] + ext[d]) / voxel) - head->qmin[d] + 1 : 1;
    }
}

// sort key = index of the point's voxel along a boustrophedon path through the voxel grid (the scan direction of an axis flips each
// time the path index of the axes before it advances): voxels that follow each other in the order are face neighbours in space, so
// ANY run of consecutive rows of the sorted cloud is spatially compact (hip.py:_serpentine did this on the cluster list)
template <typename T>
__global__ void __launch_bounds__(256) path_keys_kernel(const T* __restrict__ x, int n, int D, const SortHead* __restrict__ head,
                                                        uint64_t* __restrict__ keys, int32_t* __restrict__ idx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float voxel = head->voxel;
    uint64_t path = 0;
    for (int d = 0; d < D; ++d) {
        const int e = head->ext[d];
        int q = (int)floorf(to_f32<T>(x[i * D + d]) / voxel) - head->qmin[d];
        q = q < 0 ? 0 : (q >= e ? e - 1 : q);
        const int c = (path & 1) ? e - 1 - q : q;
        path = path * (uint64_t)e + (uint64_t)c;
    }
    keys[i] = path;
    idx[i] = (int32_t)i;
}

template <typename T>
__global__ void __launch_bounds__(256) gather_points_kernel(const T* __restrict__ x, const int32_t* __restrict__ perm, int n, int D, T* __restrict__ xs) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const long j = perm[p];
    for (int d = 0; d < D; ++d) xs[p * D + d] = x[j * D + d];
}

__global__ void __launch_bounds__(256) gather_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ perm, float* __restrict__ dst, int n) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k < n) dst[k] = src[perm[k]];
}
__global__ void __launch_bounds__(256) scatter_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ perm, float* __restrict__ dst, int n, int width) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const long j = perm[k];
    for (int d = 0; d < width; ++d) dst[j * width + d] = src[k * width + d];
}

// "every slab of kSortSlab rows x all columns, in kSortColChunks column intervals" as KeOps-style ranges
__global__ void __launch_bounds__(256) slab_ranges_kernel(int N, int M, int C, int32_t* __restrict__ ranges_i, int32_t* __restrict__ slices_i,
                                                          int32_t* __restrict__ red) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= C) return;
    ranges_i[2 * k] = k * kSortSlab;
    ranges_i[2 * k + 1] = min(N, (k + 1) * kSortSlab);
    slices_i[k] = (k + 1) * kSortColChunks;
    const int step = (((M + kSortColChunks - 1) / kSortColChunks) + 31) / 32 * 32;
    for (int c = 0; c < kSortColChunks; ++c) {
        red[2 * (k * kSortColChunks + c)] = min(M, c * step);
        red[2 * (k * kSortColChunks + c) + 1] = min(M, (c + 1) * step);
    }
}

size_t sort64_temp_bytes(int n) {
    size_t bytes = 0;
    uint64_t* k = nullptr;
    int32_t* v = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0u, 64u, (hipStream_t)0) != hipSuccess || bytes == 0)
        bytes = (size_t)n * 16 + (1 << 20);
    (void)hipGetLastError();
    return bytes;
}

template <typename T>
int compact_sort_typed(const void* z_, int n, int D, int rows_per_voxel, int32_t* perm, void* z_sorted, void* scratch, size_t scratch_bytes,
                       hipStream_t st) {
    const T* z = static_cast<const T*>(z_);
    char* ws = static_cast<char*>(scratch);
    SortHead* head = reinterpret_cast<SortHead*>(ws);
    size_t off = align256(sizeof(SortHead));
    uint64_t* keys_in = reinterpret_cast<uint64_t*>(ws + off); off += align256((size_t)n * 8);
    uint64_t* keys_out = reinterpret_cast<uint64_t*>(ws + off); off += align256((size_t)n * 8);
    int32_t* idx_in = reinterpret_cast<int32_t*>(ws + off); off += align256((size_t)n * 4);
    size_t temp = sort64_temp_bytes(n);
    if (off + temp > scratch_bytes) return fail(GLHIP_EINVAL, "compact_sort: scratch of %zu bytes, need %zu", scratch_bytes, off + temp);
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(sort_head_init_kernel, dim3(1), dim3(64), 0, st, head);
    hipLaunchKernelGGL((bbox_kernel<T>), dim3(blocks < 512 ? blocks : 512), dim3(256), 0, st, z, n, D, head);
    hipLaunchKernelGGL(voxel_kernel, dim3(1), dim3(64), 0, st, head, n, D, rows_per_voxel);
    hipLaunchKernelGGL((path_keys_kernel<T>), dim3(blocks), dim3(256), 0, st, z, n, D, head, keys_in, idx_in);
    if (rocprim::radix_sort_pairs(ws + off, temp, keys_in, keys_out, idx_in, perm, (size_t)n, 0u, 64u, st) != hipSuccess)
        return fail(GLHIP_ELAUNCH, "compact_sort: radix sort failed: %s", hipGetErrorString(hipGetLastError()));
    hipLaunchKernelGGL((gather_points_kernel<T>), dim3(blocks), dim3(256), 0, st, z, perm, n, D, static_cast<T*>(z_sorted));
    return GLHIP_OK;
}

}  // namespace

namespace glhip {

size_t compact_sort_scratch_bytes(int n) {
    return align256(sizeof(SortHead)) + 2 * align256((size_t)n * 8) + align256((size_t)n * 4) + align256(sort64_temp_bytes(n));
}

int compact_sort(const void* z, int n, int D, int in_dtype, int rows_per_voxel, int32_t* perm, void* z_sorted, void* scratch,
                 size_t scratch_bytes, hipStream_t st) {
    return in_dtype == GLHIP_F32 ? compact_sort_typed<float>(z, n, D, rows_per_voxel, perm, z_sorted, scratch, scratch_bytes, st)
                                 : compact_sort_typed<bf16_t>(z, n, D, rows_per_voxel, perm, z_sorted, scratch, scratch_bytes, st);
}

void gather_f32(const float* src, const int32_t* perm, float* dst, int n, hipStream_t st) {
    hipLaunchKernelGGL(gather_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, perm, dst, n);
}

void scatter_f32(const float* src, const int32_t* perm, float* dst, int n, hipStream_t st, int width) {
    hipLaunchKernelGGL(scatter_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, perm, dst, n, width);
}

void slab_ranges(int N, int M, int32_t* ranges_i, int32_t* slices_i, int32_t* red, hipStream_t st) {
    const int C = (N + kSortSlab - 1) / kSortSlab;
    hipLaunchKernelGGL(slab_ranges_kernel, dim3((C + 255) / 256), dim3(256), 0, st, N, M, C, ranges_i, slices_i, red);
}

}  // namespace glhip

namespace {

// ---- keep rule -> merged column intervals ------------------------------------------------------------------------------

struct KeepRule {
    int kind;              // GLHIP_KEEP_DUAL_SLACK | GLHIP_KEEP_WITHIN
    const float* rows;     // (Cr, D) centroids of the row clusters
    const float* cols;     // (Cc, D)
    const float* f;        // (Cr) dual values of the rows   (dual slack only)
    const float* g;        // (Cc)
    int D, p;
    float thr;
};

__device__ __forceinline__ bool keep_pair(const KeepRule& k, int i, int j) {
    float d2 = 0.f;
    if (k.D <= 3) {      // branch-free: the loads of several calls can be in flight together (runs_kernel)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int dd = d < k.D ? d : 0;
            const float t = k.rows[(long)i * k.D + dd] - k.cols[(long)j * k.D + dd];
            d2 = d < k.D ? __builtin_fmaf(t, t, d2) : d2;
        }
    } else {
        for (int d = 0; d < k.D; ++d) {
            const float t = k.rows[(long)i * k.D + d] - k.cols[(long)j * k.D + d];
            d2 = __builtin_fmaf(t, t, d2);
        }
    }
    if (k.kind == GLHIP_KEEP_WITHIN) return d2 <= k.thr;                       // kernel_samples.py:244-252
    const float C = (k.p == 2) ? 0.5f * d2 : sqrtf(fmaxf(d2, 1e-8f));          // cost_routines, sinkhorn_samples.py:26-29
    return k.f[i] + k.g[j] > C - k.thr;                                        // sinkhorn_samples.py:512-514
}

// One wavefront per row cluster.  A "run" is a maximal sequence of kept column clusters that are adjacent in memory
// (range end == next range start): it becomes ONE column interval (same pair set as one interval per cluster, longer
// tiles for the kernels).  Every run has one start and one stop and they alternate along the row, so the k-th start and the
// k-th stop belong to the same run: ranks are plain prefix counts.
// FILL = false: counts[i] = number of runs.  FILL = true: writes the runs at offset slices[i-1] of `red`.
template <bool FILL>
__global__ void __launch_bounds__(256) runs_kernel(KeepRule rule, int Cr, int Cc, const int32_t* __restrict__ ranges_cols,
                                                   int32_t* __restrict__ counts, const int32_t* __restrict__ slices,
                                                   int32_t* __restrict__ red, long long capacity, int32_t* overflow) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= Cr) return;
    const int base = FILL ? (i == 0 ? 0 : slices[i - 1]) : 0;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int n_starts = 0, n_stops = 0;
    int prev_keep = 0, prev_end = -1;     // column c0 - 1 (wave-uniform)
    // four 64-column steps per round: the loads behind their keep tests are independent and in flight together (a wavefront per
    // row cluster is a latency-bound walk: 27-47 us per launch over ~2000 x 2000 clusters with one step per round)
    constexpr int U = 4;
    for (int c0 = 0; c0 < Cc; c0 += 64 * U) {
        int k[U + 1], cs[U + 1], ce[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 64 * u + lane, cc = min(c, Cc - 1);      // unconditional loads, masked afterwards
            const int kk = keep_pair(rule, i, cc) ? 1 : 0, s0 = ranges_cols[2 * cc], e0 = ranges_cols[2 * cc + 1];
            k[u] = c < Cc ? kk : 0;
            cs[u] = c < Cc ? s0 : -2;
            ce[u] = c < Cc ? e0 : -3;
        }
        {   // lane 0 of the next round, for the right neighbour of the last lane
            const int c = c0 + 64 * U, cc = min(c, Cc - 1);
            const int kk = keep_pair(rule, i, cc) ? 1 : 0, s0 = ranges_cols[2 * cc];
            k[U] = c < Cc ? kk : 0;
            cs[U] = c < Cc ? s0 : -4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c0 + 64 * u >= Cc) break;      // wave-uniform
            int kl = __shfl_up(k[u], 1, 64), el = __shfl_up(ce[u], 1, 64);
            if (lane == 0) { kl = prev_keep; el = prev_end; }
            int kr = __shfl_down(k[u], 1, 64), sr = __shfl_down(cs[u], 1, 64);
            const int kn = __shfl(k[u + 1], 0, 64), sn = __shfl(cs[u + 1], 0, 64);     // first column of the next step
            if (lane == 63) { kr = kn; sr = sn; }
            const bool start = k[u] && !(kl && el == cs[u]);
            const bool stop = k[u] && !(kr && sr == ce[u]);
            const unsigned long long ms = __ballot(start), me = __ballot(stop);
            if (FILL) {
                if (start) {
                    const int slot = base + n_starts + __popcll(ms & below);
                    if (slot < capacity) red[2 * slot] = cs[u]; else *overflow = 1;
                }
                if (stop) {
                    const int slot = base + n_stops + __popcll(me & below);
                    if (slot < capacity) red[2 * slot + 1] = ce[u]; else *overflow = 1;
                }
            }
            n_starts += __popcll(ms);
            n_stops += __popcll(me);
            prev_keep = __shfl(k[u], 63, 64);
            prev_end = __shfl(ce[u], 63, 64);
        }
    }
    if (!FILL && lane == 0) counts[i] = n_starts;
}

// Pairs of POINTS a keep rule retains, without building its intervals: a wavefront per (row cluster, slab of 256 column clusters)
// adds (rows of the cluster) x (columns of the kept column clusters).  Round 5: a fixed number of workgroups walks the items and
// adds its total with ONE atomic — with one workgroup per 4 row clusters the ~500 same-address atomics of a launch (each a trip to
// memory: the 8 L2s do not share lines) cost 40 us of a 45 us launch, a 2-D grid with 3000 of them 57 us.
constexpr int kKeptBlocks = 512, kKeptWaves = 16;      // ~15 ns per atomic against ~1 us per item of a wavefront's walk: 2000 x 2000 clusters -> 2 items each

__global__ void kept_zero_kernel(unsigned long long* kept) {
    if (threadIdx.x < 3) kept[threadIdx.x] = 0;
}

// Small fills and copies as kernels of this library: the FIRST hipMemsetAsync / hipMemcpyAsync of a process loads the runtime's own
// blit kernels — 110 ms inside the first glhip_block_ranges call of a two-scale loss (round 6, tools/first_call.py)
__global__ void fill_i32_kernel(int32_t* dst, int n, int32_t v) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = v;
}
__global__ void gather2_i32_kernel(int32_t* dst, const int32_t* a, const int32_t* b) {
    if (threadIdx.x == 0) { dst[0] = *a; dst[1] = *b; }
}

__global__ void __launch_bounds__(64 * kKeptWaves) kept_pairs_kernel(KeepRule rule, int Cr, int Cc, const int32_t* __restrict__ ranges_rows,
                                                                    const int32_t* __restrict__ ranges_cols, unsigned long long* __restrict__ kept) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_slabs = (Cc + 255) / 256;
    const long n_items = (long)Cr * n_slabs;
    unsigned long long acc = 0;
    for (long item = (long)blockIdx.x * kKeptWaves + wave; item < n_items; item += (long)gridDim.x * kKeptWaves) {
        const int i = (int)(item / n_slabs), c0 = (int)(item - (long)i * n_slabs) * 256;
        long long cols = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 64 * u + lane, cc = min(c, Cc - 1);      // unconditional loads, masked afterwards
            const int len = ranges_cols[2 * cc + 1] - ranges_cols[2 * cc];
            cols += (keep_pair(rule, i, cc) && c < Cc) ? len : 0;
        }
        acc += (unsigned long long)cols * (unsigned long long)(ranges_rows[2 * i + 1] - ranges_rows[2 * i]);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ unsigned long long part[kKeptWaves];
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kKeptWaves; ++w) t += part[w];
        if (t) atomicAdd(kept, t);
    }
    if (blockIdx.x == 0 && wave < 2) {      // the sums of squared cluster sizes, rows (wavefront 0) and columns (wavefront 1), once
        const int32_t* rg = wave == 0 ? ranges_rows : ranges_cols;
        const int C = wave == 0 ? Cr : Cc;
        unsigned long long sq = 0;
        for (int c = lane; c < C; c += 64) {
            const unsigned long long m = (unsigned long long)(rg[2 * c + 1] - rg[2 * c]);
            sq += m * m;
        }
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        if (lane == 0) kept[1 + wave] = sq;
    }
}

// inclusive scan of `counts` (n <= a few 1e5) by a single workgroup -> CSR end offsets
__global__ void __launch_bounds__(1024) slices_kernel(const int32_t* __restrict__ counts, int n, int32_t* __restrict__ slices) {
    __shared__ int scan[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int k = base + tid;
        scan[tid] = k < n ? counts[k] : 0;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int v = (tid >= off) ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        if (k < n) slices[k] = carry + scan[tid];
        __syncthreads();
        if (tid == 1023) carry += scan[1023];
        __syncthreads();
    }
}

}  // namespace

extern "C" {

size_t glhip_cluster_workspace_bytes(int N, int D) {
    if (N <= 0 || D < 1 || D > 3) return 0;
    const size_t ts = sort_temp_bytes(N), tc = scan_temp_bytes(N);
    return align256(sizeof(ClusterHead)) + 2 * align256((size_t)N * 8) + 2 * align256((size_t)N * 4) + align256(ts > tc ? ts : tc);
}

int glhip_grid_cluster(const void* x, const float* weights, int N, int D, int in_dtype, float pre_div, float voxel, int32_t* perm,
                       void* x_sorted, float* w_sorted, int32_t* ranges, float* centroids, float* weights_c, int32_t* n_clusters,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (N < 0 || D < 1 || D > 3) return fail(N < 0 ? GLHIP_EINVAL : GLHIP_EUNSUPPORTED, "glhip_grid_cluster: bad sizes N=%d D=%d (D <= 3)", N, D);
    if (in_dtype != GLHIP_F32 && in_dtype != GLHIP_BF16) return fail(GLHIP_EINVAL, "glhip_grid_cluster: bad in_dtype %d", in_dtype);
    if (!(voxel > 0.f) || !(pre_div > 0.f)) return fail(GLHIP_EINVAL, "glhip_grid_cluster: voxel and pre_div must be > 0");
    if (!n_clusters) return fail(GLHIP_EINVAL, "glhip_grid_cluster: NULL n_clusters");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (N == 0) {
        hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, st, n_clusters, 8, 0);
        return GLHIP_OK;
    }
    if (!x || !perm || !ranges || !centroids || !weights_c || !workspace)
        return fail(GLHIP_EINVAL, "glhip_grid_cluster: NULL pointer");
    return in_dtype == GLHIP_F32
               ? grid_cluster_typed<float>(x, weights, N, D, pre_div, voxel, perm, x_sorted, w_sorted, ranges, centroids, weights_c, n_clusters, workspace, workspace_bytes, st)
               : grid_cluster_typed<bf16_t>(x, weights, N, D, pre_div, voxel, perm, x_sorted, w_sorted, ranges, centroids, weights_c, n_clusters, workspace, workspace_bytes, st);
}

namespace {
int check_block_ranges(const char* fn, int kind, const float* rows, const float* cols, const float* f, const float* g, int Cr, int Cc, int D,
                       int p, const int32_t* ranges_rows, const int32_t* ranges_cols, const int32_t* slices_rows, const int32_t* slices_cols) {
    if (kind != GLHIP_KEEP_DUAL_SLACK && kind != GLHIP_KEEP_WITHIN) return fail(GLHIP_EINVAL, "%s: bad kind %d", fn, kind);
    if (Cr < 0 || Cc < 0 || D < 1) return fail(GLHIP_EINVAL, "%s: bad sizes", fn);
    if (kind == GLHIP_KEEP_DUAL_SLACK && p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "%s: p must be 1 or 2", fn);
    if (Cr == 0 || Cc == 0) return GLHIP_OK;
    if (!rows || !cols || !ranges_rows || !ranges_cols || !slices_rows || !slices_cols || (kind == GLHIP_KEEP_DUAL_SLACK && (!f || !g)))
        return fail(GLHIP_EINVAL, "%s: NULL pointer", fn);
    return GLHIP_OK;
}
}  // namespace

int glhip_block_ranges_count(int kind, const float* rows, const float* cols, const float* f, const float* g, int Cr, int Cc, int D, int p,
                             float thr, const int32_t* ranges_rows, const int32_t* ranges_cols, int32_t* slices_rows,
                             int32_t* slices_cols, int32_t* totals, void* stream) {
    const int rc = check_block_ranges("glhip_block_ranges_count", kind, rows, cols, f, g, Cr, Cc, D, p, ranges_rows, ranges_cols, slices_rows, slices_cols);
    if (rc) return rc;
    if (!totals) return fail(GLHIP_EINVAL, "glhip_block_ranges_count: NULL totals");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (Cr == 0 || Cc == 0) {
        hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, st, totals, 2, 0);
        return GLHIP_OK;
    }
    const KeepRule fwd{kind, rows, cols, f, g, D, p, thr}, bwd{kind, cols, rows, g, f, D, p, thr};
    hipLaunchKernelGGL((runs_kernel<false>), dim3((Cr + 3) / 4), dim3(256), 0, st, fwd, Cr, Cc, ranges_cols, slices_rows, nullptr, nullptr, 0LL, nullptr);
    hipLaunchKernelGGL(slices_kernel, dim3(1), dim3(1024), 0, st, slices_rows, Cr, slices_rows);
    hipLaunchKernelGGL((runs_kernel<false>), dim3((Cc + 3) / 4), dim3(256), 0, st, bwd, Cc, Cr, ranges_rows, slices_cols, nullptr, nullptr, 0LL, nullptr);
    hipLaunchKernelGGL(slices_kernel, dim3(1), dim3(1024), 0, st, slices_cols, Cc, slices_cols);
    hipLaunchKernelGGL(gather2_i32_kernel, dim3(1), dim3(64), 0, st, totals, slices_rows + (Cr - 1), slices_cols + (Cc - 1));
    return check_launch("glhip_block_ranges_count");
}

int glhip_block_ranges(int kind, const float* rows, const float* cols, const float* f, const float* g, int Cr, int Cc, int D, int p,
                       float thr, const int32_t* ranges_rows, const int32_t* ranges_cols, int32_t* slices_rows, int32_t* red_cols,
                       int32_t* slices_cols, int32_t* red_rows, long long capacity, int32_t* status, void* stream) {
    // slices_cols == red_rows == NULL: the row-major pattern only (a caller whose pattern is symmetric — the debiasing terms of a
    // Sinkhorn divergence: rows = cols, f = g — uses it for both orientations: two launches of the rule and one scan less)
    const bool both = slices_cols != nullptr || red_rows != nullptr;
    const int rc = check_block_ranges("glhip_block_ranges", kind, rows, cols, f, g, Cr, Cc, D, p, ranges_rows, ranges_cols, slices_rows,
                                      both ? slices_cols : slices_rows);
    if (rc) return rc;
    if (capacity < 0) return fail(GLHIP_EINVAL, "glhip_block_ranges: capacity < 0");
    if (Cr == 0 || Cc == 0) return GLHIP_OK;
    if (!red_cols || (both && !red_rows) || !status) return fail(GLHIP_EINVAL, "glhip_block_ranges: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, st, status, 1, 0);
    const KeepRule fwd{kind, rows, cols, f, g, D, p, thr}, bwd{kind, cols, rows, g, f, D, p, thr};
    // the CSR offsets double as the count buffers of the first pass
    hipLaunchKernelGGL((runs_kernel<false>), dim3((Cr + 3) / 4), dim3(256), 0, st, fwd, Cr, Cc, ranges_cols, slices_rows, nullptr, nullptr, 0LL, status);
    hipLaunchKernelGGL(slices_kernel, dim3(1), dim3(1024), 0, st, slices_rows, Cr, slices_rows);
    hipLaunchKernelGGL((runs_kernel<true>), dim3((Cr + 3) / 4), dim3(256), 0, st, fwd, Cr, Cc, ranges_cols, nullptr, slices_rows, red_cols, capacity, status);
    if (both) {
        hipLaunchKernelGGL((runs_kernel<false>), dim3((Cc + 3) / 4), dim3(256), 0, st, bwd, Cc, Cr, ranges_rows, slices_cols, nullptr, nullptr, 0LL, status);
        hipLaunchKernelGGL(slices_kernel, dim3(1), dim3(1024), 0, st, slices_cols, Cc, slices_cols);
        hipLaunchKernelGGL((runs_kernel<true>), dim3((Cc + 3) / 4), dim3(256), 0, st, bwd, Cc, Cr, ranges_rows, nullptr, slices_cols, red_rows, capacity, status);
    }
    return check_launch("glhip_block_ranges");
}

int glhip_block_ranges_kept_pairs(int kind, const float* rows, const float* cols, const float* f, const float* g, int Cr, int Cc, int D,
                                  int p, float thr, const int32_t* ranges_rows, const int32_t* ranges_cols, long long* kept, void* stream) {
    if (kind != GLHIP_KEEP_DUAL_SLACK && kind != GLHIP_KEEP_WITHIN) return fail(GLHIP_EINVAL, "glhip_block_ranges_kept_pairs: bad kind %d", kind);
    if (Cr < 0 || Cc < 0 || D < 1) return fail(GLHIP_EINVAL, "glhip_block_ranges_kept_pairs: bad sizes");
    if (kind == GLHIP_KEEP_DUAL_SLACK && p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "glhip_block_ranges_kept_pairs: p must be 1 or 2");
    if (!kept) return fail(GLHIP_EINVAL, "glhip_block_ranges_kept_pairs: NULL kept");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(kept_zero_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned long long*>(kept));      // (a 24-byte hipMemsetAsync is two fill kernels)
    if (Cr == 0 || Cc == 0) return GLHIP_OK;
    if (!rows || !cols || !ranges_rows || !ranges_cols || (kind == GLHIP_KEEP_DUAL_SLACK && (!f || !g)))
        return fail(GLHIP_EINVAL, "glhip_block_ranges_kept_pairs: NULL pointer");
    const KeepRule rule{kind, rows, cols, f, g, D, p, thr};
    const long items = (long)Cr * ((Cc + 255) / 256);
    const long want = (items + kKeptWaves - 1) / kKeptWaves;
    hipLaunchKernelGGL(kept_pairs_kernel, dim3(want < kKeptBlocks ? (unsigned)want : kKeptBlocks), dim3(64 * kKeptWaves), 0, st, rule, Cr, Cc, ranges_rows, ranges_cols,
                       reinterpret_cast<unsigned long long*>(kept));
    return check_launch("glhip_block_ranges_kept_pairs");
}

}  // extern "C"
