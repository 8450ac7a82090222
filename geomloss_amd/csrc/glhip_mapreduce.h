// glhip_mapreduce.h — the one tiling skeleton shared by every low-dimensional (D <= 3)
// reduction of this library:  out_i = REDUCE_j F(x_i, y_j, s_j).
//
// Work decomposition (CDNA4):
//   * one workgroup = 256 threads = 4 wavefronts; each thread owns R rows i, held in VGPRs;
//     rows are assigned lane-contiguously (row = base + r*256 + tid) so x loads / out stores coalesce.
//   * the column cloud is streamed through LDS in tiles of kTile records of 16 B
//     ({y_0, y_1, y_2, s_j}); every lane of a wavefront reads the same record at the same time,
//     i.e. one conflict-free broadcast ds_read_b128 per column and per R rows.
//   * dense mode: grid.x tiles the rows, grid.y = batch.  Block-sparse mode: grid.x = row block k of
//     the KeOps-style ranges; the workgroup walks the column intervals of its CSR slice.
//   * coordinates are re-centred on the first row of the workgroup before any product is formed,
//     so the expanded form  -|x-y|^2/2 = x.y - |x|^2/2 - |y|^2/2  is evaluated on offsets that are
//     at most one cloud diameter long (and much shorter when the rows are cluster-sorted).
//
// An `Op` supplies: Params, RowState, kRows (R), kDim (D), and the device functions
//   init_rows / make_record / neutral_record / consume / finish_rows.
#pragma once

#include "glhip_common.h"

namespace glhip {

template <class Op, bool SPARSE>
__global__ void __launch_bounds__(kBlock)
mapreduce_kernel(typename Op::Params prm, Ranges rg, int N, int M) {
    constexpr int D = Op::kDim;
    constexpr int R = Op::kRows;
    constexpr int kRowsPerPass = kBlock * R;
    __shared__ Rec<D> tile[kTile];

    const int tid = threadIdx.x;
    const int b = blockIdx.y;

    int row_begin, row_end, q_begin, q_end;
    if (SPARSE) {
        const int k = blockIdx.x;
        row_begin = rg.ranges_i[2 * k];
        row_end = rg.ranges_i[2 * k + 1];
        q_begin = (k == 0) ? 0 : rg.slices_i[k - 1];
        q_end = rg.slices_i[k];
    } else {
        row_begin = blockIdx.x * kRowsPerPass;
        row_end = min(N, row_begin + kRowsPerPass);
        q_begin = 0;
        q_end = 1;
    }

    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerPass) {
        // centre of this pass: its first row (wave-uniform, read through the scalar path)
        float centre[D];
        Op::load_centre(prm, b, N, row0, centre);

        typename Op::RowState st;
        Op::init_rows(prm, b, N, row0, row_end, tid, centre, st);

        for (int q = q_begin; q < q_end; ++q) {
            const int js = SPARSE ? rg.redranges_j[2 * q] : 0;
            const int je = SPARSE ? rg.redranges_j[2 * q + 1] : M;
            for (int j0 = js; j0 < je; j0 += kTile) {
                const int n = min(kTile, je - j0);
                const int npad = (n + kChunk - 1) & ~(kChunk - 1);
                __syncthreads();   // previous tile fully consumed
                for (int t = tid; t < npad; t += kBlock) {
                    tile[t] = (t < n) ? Op::make_record(prm, b, M, j0 + t, centre)
                                      : Op::neutral_record();
                }
                __syncthreads();
                for (int jj = 0; jj < npad; jj += kChunk) Op::consume(st, &tile[jj]);
            }
        }
        Op::finish_rows(prm, b, N, row0, row_end, tid, centre, st);
    }
}

// Host-side launch helper.
template <class Op>
static inline void launch_mapreduce(const typename Op::Params& prm, const Ranges& rg, int n_ranges,
                                    int B, int N, int M, hipStream_t stream) {
    if (n_ranges > 0) {
        dim3 grid(n_ranges, 1, 1);
        hipLaunchKernelGGL((mapreduce_kernel<Op, true>), grid, dim3(kBlock), 0, stream, prm, rg, N, M);
    } else {
        const int rows_per_block = kBlock * Op::kRows;
        dim3 grid((N + rows_per_block - 1) / rows_per_block, B, 1);
        hipLaunchKernelGGL((mapreduce_kernel<Op, false>), grid, dim3(kBlock), 0, stream, prm, rg, N, M);
    }
}

}  // namespace glhip
