// glhip_mapreduce.h — the tiling skeleton shared by the low-dimensional (D <= 3) reductions
//   out_i = REDUCE_j F(x_i, y_j, s_j).
//
// Work decomposition (CDNA4):
//   * one workgroup = 256 threads = 4 wavefronts; each thread owns R rows i, held in VGPRs;
//     rows are assigned lane-contiguously (row = base + r*256 + tid) so x loads / out stores coalesce.
//   * the column cloud is streamed through LDS in tiles of kTile records of 16 B
//     ({y_0, y_1, y_2, s_j}); every lane of a wavefront reads the same record at the same time,
//     i.e. one conflict-free broadcast ds_read_b128 per column and per R rows.
//   * grid = (row blocks) x (batch) x (column splits).  A workgroup that owns all the columns of its
//     rows writes the final result; otherwise it writes a per-row partial state to the caller's
//     workspace and `merge_kernel` combines the splits.  Column splits exist for load balance: one
//     workgroup working through 1e6 columns runs for ~the whole kernel, so without them the last
//     partially filled round of workgroups leaves most CUs idle (measured: 4.2 of 7 resident waves/SIMD).
//   * dense mode: grid.x tiles the rows.  Block-sparse mode: grid.x = row block k of the KeOps-style
//     ranges; split s of a row block takes the column intervals q = s (mod n_splits) of its CSR slice.
//   * coordinates are re-centred on the first row of the workgroup's row pass before any product is
//     formed, so the expanded form  -|x-y|^2/2 = x.y - |x|^2/2 - |y|^2/2  is evaluated on offsets that
//     are at most one cloud diameter long (and much shorter when the rows are cluster-sorted).
//
// An `Op` supplies: Params, RowState, kRows (R), kDim (D), kPartial and the device functions
//   load_centre / init_rows / make_record / neutral_record / consume / finish_rows /
//   store_partial / merge_row.
#pragma once

#include "glhip_common.h"

namespace glhip {

struct SplitInfo {
    int n_splits;       // 1 = no split
    float* workspace;   // [n_splits][B*N][kPartial] floats
    long split_stride;  // floats between consecutive splits = B*N*kPartial
    int xcd_grid_x;     // > 0: 1-D XCD-aware grid (see workgroup_coords); value = number of row blocks per batch item
    int xcd_blocks;     // XCD-aware grid: row blocks x batch items (workgroups per column split)
    int gather = 0;     // block-sparse forward kernel: tiles gather several short column intervals (glhip_softmin_x32.h: gather_tile)
    int share = 0;      // block-sparse forward kernel, 4 wavefronts: the chunk table was built in share mode (build_row_chunks_kernel)
};

// Logical (row block, batch item, column split) of this workgroup.  Plain mode: the 3-D grid.  XCD-aware mode
// (dense kernels with n_splits == 8): a 1-D grid whose linear id is split-major modulo 8.  The dispatcher places
// workgroup b on XCD b % 8 (observed behaviour, used for speed only), so XCD k only ever streams the k-th eighth
// of the column cloud: at M = 1e6 that is 2 MB of points + dual values, resident in the XCD's 4 MB L2 instead of
// being re-fetched through the fabric by every row block.
__device__ __forceinline__ void workgroup_coords(const SplitInfo& sp, int& bx, int& by, int& bz) {
    if (sp.xcd_grid_x > 0) {
        // linear id = ((phase * blocks + block) * 8 + xcd): XCD k runs split k of every row block first, then split
        // k + 8, ... — at any time its L2 holds one split's columns, however many splits there are
        const int lid = blockIdx.x;
        const int xcd = lid & 7;
        const int t = lid >> 3;
        const int phase = t / sp.xcd_blocks;
        const int block = t - phase * sp.xcd_blocks;
        bz = phase * 8 + xcd;
        bx = block % sp.xcd_grid_x;
        by = block / sp.xcd_grid_x;
    } else {
        bx = blockIdx.x;
        by = blockIdx.y;
        bz = blockIdx.z;
    }
}

// rows [row_begin,row_end) and CSR slice [q_begin,q_end) of workgroup blockIdx.x
template <bool SPARSE>
__device__ __forceinline__ void block_extent(const Ranges& rg, int N, int rows_per_pass, int& row_begin,
                                             int& row_end, int& q_begin, int& q_end, int bx = blockIdx.x) {
    if (SPARSE) {
        int k = bx;
        const int T = rg.chunks ? rg.chunks[0] : 0;
        if (rg.chunks && T >= 0) {   // row blocks cut into chunks of at most `rows_per_pass` rows: workgroup bx owns chunk bx
            if (bx >= T) { row_begin = row_end = q_begin = q_end = 0; return; }
            k = rg.chunks[1 + 3 * bx];
            row_begin = rg.chunks[2 + 3 * bx];
            row_end = rg.chunks[3 + 3 * bx];
        } else if (rg.chunks) {      // the table overflowed (T = -n_ranges): one workgroup per row block, the KeOps granularity
            if (bx >= -T) { row_begin = row_end = q_begin = q_end = 0; return; }
            row_begin = rg.ranges_i[2 * k];
            row_end = rg.ranges_i[2 * k + 1];
        } else {
            row_begin = rg.ranges_i[2 * k];
            row_end = rg.ranges_i[2 * k + 1];
        }
        q_begin = (k == 0) ? 0 : rg.slices_i[k - 1];
        q_end = rg.slices_i[k];
    } else {
        row_begin = bx * rows_per_pass;
        row_end = min(N, row_begin + rows_per_pass);
        q_begin = 0;
        q_end = 1;
    }
}

// column interval q as seen by split `s` of `ns`
template <bool SPARSE>
__device__ __forceinline__ void column_interval(const Ranges& rg, int M, int q, int s, int ns, int& js, int& je) {
    if (SPARSE) {
        js = rg.redranges_j[2 * q];
        je = rg.redranges_j[2 * q + 1];
    } else {
        const int len = (((M + ns - 1) / ns) + kChunk - 1) & ~(kChunk - 1);
        js = min(M, s * len);
        je = min(M, js + len);
    }
}

// Row-chunk table of a block-sparse launch.  The KeOps convention gives one workgroup per row block (cluster); voxel clusters
// of real clouds are very uneven (a sphere sampled at 1e6 points: 1 ... 19 566 rows per cluster), so the workgroup of the
// biggest cluster runs long after the chip has drained.  This single-workgroup kernel cuts every row block into chunks of
// `rows` rows (the row tile of the kernel that follows) with a block-wide prefix sum over the clusters, so that the grid of
// the reduction is one workgroup per chunk: chunks[0] = T, chunks[1 + 3c ...] = (row block, first row, end row).  The host
// only knows the bound T <= n_ranges + N / rows (valid for disjoint row blocks) and launches that many workgroups; the surplus
// exits at once.  chunks[0] < 0 = "table too small, ignore it" (see the end of the kernel).
static __global__ void __launch_bounds__(1024)
build_row_chunks_kernel(const int32_t* __restrict__ ranges_i, int n_ranges, int rows, int32_t* __restrict__ chunks, int capacity,
                        int share = 0) {
    __shared__ int scan[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_ranges; base += 1024) {
        const int k = base + tid;
        int r0 = 0, r1 = 0;
        if (k < n_ranges) { r0 = ranges_i[2 * k]; r1 = ranges_i[2 * k + 1]; }
        int cnt = (r1 > r0) ? (r1 - r0 + rows - 1) / rows : 0;
        // share (rows = 128 = 4 row tiles of 32; glhip_softmin_x32.h): a row block of nt > 4 row tiles gets W = nt / 4 chunks of exactly
        // 4 tiles; of the rl = nt mod 4 tiles left over, min(rl, W) are reduced on the side by the workgroups of the first chunks
        // (one each) and belong to no chunk; the others (W < rl: blocks of 5 - 7, 11 row tiles) form a trailing partial chunk
        int full = cnt, skip = 0;      // chunks of `rows` rows; share: rows of the carried tiles, between the last full chunk and the trailing one
        if (share && cnt > 1) {
            const int nt = (r1 - r0 + 31) >> 5, W = nt >> 2, rl = nt - 4 * W, e = rl < W ? rl : W;
            full = W;
            skip = 32 * e;
            cnt = W + (rl > e ? 1 : 0);
        }
        scan[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {   // inclusive Hillis-Steele scan
            const int v = (tid >= off) ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const int first = carry + scan[tid] - cnt;
        for (int c = 0; c < cnt; ++c) {
            const int slot = first + c;
            if (slot < capacity) {
                chunks[1 + 3 * slot] = k;
                chunks[2 + 3 * slot] = r0 + c * rows + (c < full ? 0 : skip);
                chunks[3 + 3 * slot] = c < full ? min(r1, r0 + (c + 1) * rows) : r1;
            }
        }
        __syncthreads();
        if (tid == 1023) carry += scan[1023];
        __syncthreads();
    }
    // The host sizes the table for DISJOINT row blocks inside [0, N) (n_ranges + N / rows chunks always suffice then).  Overlapping
    // or out-of-range blocks can need more: instead of dropping the surplus chunks, tell the reduction to ignore the table and run
    // one workgroup per row block (the grid, >= n_ranges workgroups, covers that too).
    if (tid == 0) chunks[0] = (carry <= capacity) ? carry : -n_ranges;
}

template <class Op, bool SPARSE>
__global__ void __launch_bounds__(kBlock)
mapreduce_kernel(typename Op::Params prm, Ranges rg, int N, int M, SplitInfo sp) {
    constexpr int D = Op::kDim;
    constexpr int R = Op::kRows;
    constexpr int kRowsPerPass = kBlock * R;
    __shared__ Rec<D> tile[kTile];

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int split = blockIdx.z;
    const int ns = sp.n_splits;

    int row_begin, row_end, q_begin, q_end;
    block_extent<SPARSE>(rg, N, kRowsPerPass, row_begin, row_end, q_begin, q_end);

    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerPass) {
        float centre[D];
        Op::load_centre(prm, b, N, row0, centre);   // wave-uniform (scalar loads)

        typename Op::RowState st;
        Op::init_rows(prm, b, N, row0, row_end, tid, centre, st);

        for (int q = q_begin + (SPARSE ? split : 0); q < q_end; q += (SPARSE ? ns : 1)) {
            int js, je;
            column_interval<SPARSE>(rg, M, q, split, ns, js, je);
            for (int j0 = js; j0 < je; j0 += kTile) {
                const int n = min(kTile, je - j0);
                const int npad = (n + kChunk - 1) & ~(kChunk - 1);
                __syncthreads();   // previous tile fully consumed
                for (int t = tid; t < npad; t += kBlock) {
                    tile[t] = (t < n) ? Op::make_record(prm, b, M, j0 + t, centre) : Op::neutral_record();
                }
                __syncthreads();
                for (int jj = 0; jj < npad; jj += kChunk) Op::consume(st, &tile[jj]);
            }
        }
        if (ns == 1) {
            Op::finish_rows(prm, b, N, row0, row_end, tid, centre, st);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = row0 + r * kBlock + tid;
                if (i < row_end)
                    Op::store_partial(st, r, sp.workspace + split * sp.split_stride +
                                                 ((long)b * N + i) * Op::kPartial);
            }
        }
    }
}

// Combines the column splits of every row: one thread per row.
template <class Op, bool SPARSE>
__global__ void __launch_bounds__(kBlock)
merge_kernel(typename Op::Params prm, Ranges rg, int N, SplitInfo sp) {
    constexpr int D = Op::kDim;
    constexpr int kRowsPerPass = kBlock * Op::kRows;
    const int b = blockIdx.y;
    int row_begin, row_end, q_begin, q_end;
    block_extent<SPARSE>(rg, N, kRowsPerPass, row_begin, row_end, q_begin, q_end);
    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerPass) {
        float centre[D];
        Op::load_centre(prm, b, N, row0, centre);
        for (int i = row0 + threadIdx.x; i < min(row_end, row0 + kRowsPerPass); i += kBlock) {
            Op::merge_row(prm, b, N, i, centre, sp.workspace + ((long)b * N + i) * Op::kPartial, sp.n_splits,
                          sp.split_stride);
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------

// Number of column splits: enough workgroups for >= ~16 rounds over the chip.
static inline int choose_splits(long row_blocks, int M, int n_ranges, long max_by_workspace) {
    // dense launches that already fill the chip four times over gain nothing from splits and pay one prologue / epilogue / merge share
    // per split: B = 256 x 4096 x 4096 (4096 row blocks, 3 splits by the rule below): 17.4 -> 15.9 ms per loss without (round 4)
    // (short column loops only: with long ones the last partial round of long-lived workgroups is what splits are for)
    if (n_ranges == 0 && row_blocks >= 4L * 768 && M <= 16384) return 1;
    const long target = 256L * 4 * 10;   // ~10 rounds of 4 workgroups per CU
    long ns = (target + row_blocks - 1) / row_blocks;
    const long by_cols = (n_ranges > 0) ? 8 : (long)M / 512;   // at least 512 columns per split
    ns = ns < by_cols ? ns : by_cols;
    ns = ns < 32 ? ns : 32;
    ns = ns < max_by_workspace ? ns : max_by_workspace;
    return ns < 1 ? 1 : (int)ns;
}

// XCD-aware dense launches use a multiple of 8 column splits (split s runs on XCD s % 8).  All workgroups take the
// same time, so the kernel lasts ceil(workgroups / resident slots) rounds: with 8 splits a 1e5-row problem is 4.07
// rounds of work spread over 5 (81 %).  Take the smallest multiple of 8 (up to 32) that fills its last round to
// >= 93 %, else the best one.  `slots` = workgroups resident on the chip (256 CUs x workgroups per CU).
static inline int xcd_splits(long row_blocks, int M, long slots, long max_by_workspace) {
    int best = 8;
    double best_eff = 0.0;
    for (int ns = 8; ns <= 32; ns += 8) {
        if (ns > max_by_workspace || (long)M / ns < 2048) break;
        const double w = (double)row_blocks * ns / (double)slots;
        const double rounds = (double)((row_blocks * ns + slots - 1) / slots);
        const double eff = w / rounds;
        if (eff >= 0.93) return ns;
        if (eff > best_eff + 0.02) { best_eff = eff; best = ns; }
    }
    return best;
}

// Mid-size dense launches whose packed columns fit the L2 of EVERY XCD at once (<= 3.5 MB: M = 1e5 at 32 bytes per column) need no
// XCD-aware placement, so the number of splits need not be a multiple of 8: the smallest ns in [4, 32] (>= 2048 columns per split)
// that fills its last round of resident workgroups to >= 97 %, else the best.  N = M = 1e5, 196 row blocks of 512 rows on 512
// slots: ns = 32 is 12.25 rounds spread over 13 (0.94) with 6 tiles per workgroup; ns = 13 is 4.98 rounds over 5 with 15 tiles.
static inline int free_splits(long row_blocks, int M, long slots, long max_by_workspace, double* eff_out) {
    int best = 0;
    double best_eff = 0.0;
    for (int ns = 4; ns <= 32; ++ns) {
        if (ns > max_by_workspace || (long)M / ns < 2048) break;
        const double w = (double)row_blocks * ns / (double)slots;
        const double eff = w / (double)((row_blocks * ns + slots - 1) / slots);
        if (eff >= 0.97) { best = ns; best_eff = eff; break; }
        if (eff > best_eff + 0.01) { best_eff = eff; best = ns; }
    }
    if (eff_out) *eff_out = best_eff;
    return best;
}

// ... and when the columns are pre-packed (64 bytes each), enough splits for one split's records to stay resident in the
// 4 MB L2 of the XCD that streams them (workgroup_coords runs one split per XCD at a time).
static inline int xcd_splits_prepacked(long row_blocks, int M, long slots, long max_by_workspace, double bytes_per_column = 64.0) {
    int ns = xcd_splits(row_blocks, M, slots, max_by_workspace);
    while (ns + 8 <= 32 && ns + 8 <= max_by_workspace && (double)M / ns * bytes_per_column > 2.5e6) ns += 8;
    return ns;
}

// Space reserved at the front of the caller's workspace for the row-chunk table of a block-sparse launch.
struct ChunkBuf {
    int32_t* buf = nullptr;
    long capacity = 0;      // chunks the table can hold
};

static inline size_t chunk_table_bytes(int n_ranges, int N, int rows) {
    return (((size_t)(1 + 3 * ((size_t)n_ranges + (size_t)N / rows + 1)) * sizeof(int32_t)) + 255) & ~(size_t)255;
}

// Queues build_row_chunks_kernel for a reduction whose workgroups take `rows` rows and returns the ranges to hand to it
// together with its grid.x; without a reserved table (no workspace) the launch keeps one workgroup per row block.
static inline Ranges with_row_chunks(const Ranges& rg, int n_ranges, int N, int rows, const ChunkBuf& cb, hipStream_t stream,
                                     unsigned& grid_x, int share = 0) {
    grid_x = (unsigned)n_ranges;
    const long bound = (long)n_ranges + N / rows;
    if (!cb.buf || cb.capacity < bound || n_ranges <= 0) return rg;
    hipLaunchKernelGGL(build_row_chunks_kernel, dim3(1), dim3(1024), 0, stream, rg.ranges_i, n_ranges, rows, cb.buf, (int)bound, share);
    Ranges out = rg;
    out.chunks = cb.buf;
    grid_x = (unsigned)bound;
    return out;
}

template <class Op>
static inline void launch_mapreduce(const typename Op::Params& prm, const Ranges& rg, int n_ranges, int B, int N,
                                    int M, void* workspace, size_t workspace_bytes, bool allow_split,
                                    hipStream_t stream, const ChunkBuf& cb = ChunkBuf()) {
    const int rows_per_block = kBlock * Op::kRows;
    unsigned chunk_grid = 0;
    const Ranges rgc = n_ranges > 0 ? with_row_chunks(rg, n_ranges, N, rows_per_block, cb, stream, chunk_grid) : rg;
    const long row_blocks = n_ranges > 0 ? (long)n_ranges : (long)B * ((N + rows_per_block - 1) / rows_per_block);
    const long per_split = (long)B * N * Op::kPartial * sizeof(float);
    const long fit = (workspace && per_split > 0) ? (long)(workspace_bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (allow_split && fit >= 2) ? choose_splits(row_blocks, M, n_ranges, fit) : 1;
    sp.workspace = static_cast<float*>(workspace);
    sp.split_stride = (long)B * N * Op::kPartial;
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    if (n_ranges > 0) {
        dim3 grid(chunk_grid, 1, sp.n_splits);
        hipLaunchKernelGGL((mapreduce_kernel<Op, true>), grid, dim3(kBlock), 0, stream, prm, rgc, N, M, sp);
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<Op, true>), dim3(n_ranges, 1, 1), dim3(kBlock), 0, stream, prm, rg, N, sp);
    } else {
        dim3 grid((N + rows_per_block - 1) / rows_per_block, B, sp.n_splits);
        hipLaunchKernelGGL((mapreduce_kernel<Op, false>), grid, dim3(kBlock), 0, stream, prm, rg, N, M, sp);
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<Op, false>), dim3(grid.x, B, 1), dim3(kBlock), 0, stream, prm, rg, N, sp);
    }
}

}  // namespace glhip
