// glhip_launch.h — host side shared by the translation units of libgeomloss_hip.so: argument checks, scratch layout, kernel
// selection and launch templates.  Everything lives in an anonymous namespace on purpose: each .hip file includes this header and
// instantiates only the templates its entry points use (forward / gradient / kernel-product kernels compile in parallel).
#pragma once

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "glhip_error.h"
#include "glhip_generic.h"
#include "glhip_kconv_ops.h"
#include "glhip_softmin_ops.h"
#include "glhip_softmin_mfma.h"
#include "glhip_wsum_mfma.h"
#include "glhip_softmin_xdl.h"
#include "glhip_softmin_x32.h"
#include "glhip_wsum_x32.h"
#include "glhip_dist_x32.h"
#include "glhip_dist_grad_x32.h"
#include "glhip_softmin_xd.h"
#include "glhip_wsum_t32.h"
#include "glhip_dist_xd.h"

using namespace glhip;

namespace {

int check_common(const char* fn, const void* x, const void* y, const void* s, int B, int N, int M, int D,
                 int in_dtype, const int32_t* ri, const int32_t* si, const int32_t* rj, int n_ranges) {
    if (B < 0 || N < 0 || M < 0 || D < 1) return fail(GLHIP_EINVAL, "%s: bad sizes B=%d N=%d M=%d D=%d", fn, B, N, M, D);
    // empty clouds may come with NULL pointers (a torch tensor with no elements has none)
    if ((!x && (long)B * N > 0) || ((!y || !s) && (long)B * M > 0)) return fail(GLHIP_EINVAL, "%s: NULL input pointer", fn);
    if (in_dtype != GLHIP_F32 && in_dtype != GLHIP_BF16) return fail(GLHIP_EINVAL, "%s: bad in_dtype %d", fn, in_dtype);
    if (n_ranges < 0) return fail(GLHIP_EINVAL, "%s: n_ranges < 0", fn);
    if (n_ranges > 0) {
        if (!ri || !si || !rj) return fail(GLHIP_EINVAL, "%s: block-sparse mode needs ranges_i, slices_i, redranges_j", fn);
        if (B != 1) return fail(GLHIP_EUNSUPPORTED, "%s: block-sparse mode requires B == 1 (got %d)", fn, B);
    }
    if (B > 65535) return fail(GLHIP_EUNSUPPORTED, "%s: B=%d exceeds the grid.y limit 65535", fn, B);
    return GLHIP_OK;
}

struct Scratch {
    void* ws;
    size_t bytes;
    bool allow_split;
    bool force_pre;     // GLHIP_FLAG_PREPACK
    bool small_rows;    // GLHIP_FLAG_SMALL_ROW_BLOCKS
    bool h2;            // GLHIP_FLAG_F16X2: exponents from f16 x 2 pieces (glhip_softmin_xd.h) where a kernel has that layout
    ChunkBuf cb;        // block-sparse launches: room for the row-chunk table, carved off the front of the workspace
    // pre-packed column records pay for their extra launch from ~5e8 pairs on; they live in the workspace, which
    // GLHIP_FLAG_NO_SPLIT tells us to leave alone
    bool prepack(double pairs) const { return ws && (force_pre || (allow_split && pairs >= prepack_min_pairs())); }
    static double prepack_min_pairs() {     // tuning knob: GLHIP_PREPACK_MIN (pairs per launch)
        static const double v = getenv("GLHIP_PREPACK_MIN") ? atof(getenv("GLHIP_PREPACK_MIN")) : 5e8;
        return v;
    }
};

// Scratch of one API call.  Block-sparse calls reserve the front of the workspace for the row-chunk table (sized for the
// smallest row tile of the call: 128 rows, 64 under GLHIP_FLAG_SMALL_ROW_BLOCKS); the rest serves the column splits and the packed
// columns as before.
Scratch make_scratch(void* workspace, size_t bytes, int flags, int n_ranges, int N) {
    Scratch sc{workspace, bytes, (flags & GLHIP_FLAG_NO_SPLIT) == 0, (flags & GLHIP_FLAG_PREPACK) != 0,
               (flags & GLHIP_FLAG_SMALL_ROW_BLOCKS) != 0, (flags & GLHIP_FLAG_F16X2) != 0, ChunkBuf()};
    if (n_ranges > 0 && workspace) {
        // 64-row workgroups (GLHIP_FLAG_SMALL_ROW_BLOCKS) cut a cluster into twice as many chunks: their table is sized for 64-row tiles
        // (glhip_workspace_bytes reserves that much for every block-sparse call); with less workspace, the 128-row table.
        for (int rows = sc.small_rows ? 64 : 128; rows <= 128; rows *= 2) {
            const size_t need = chunk_table_bytes(n_ranges, N, rows);
            if (bytes >= need) {
                sc.cb.buf = static_cast<int32_t*>(workspace);
                sc.cb.capacity = (long)n_ranges + N / rows + 1;
                sc.ws = static_cast<char*>(workspace) + need;
                sc.bytes = bytes - need;
                if (sc.bytes == 0) sc.ws = nullptr;
                break;
            }
        }
    }
    return sc;
}

// rows per thread: 2 keeps the LDS read rate at half a ds_read_b128 per row-column step while leaving
// enough workgroups to fill 256 CUs; small problems use 1 to expose more workgroups.
inline bool use_two_rows(int B, int N, int n_ranges, const Scratch& sc) {
    if (n_ranges > 0) return true;
    if (sc.ws && sc.allow_split) return (long)B * N >= 4 * kBlock;   // column splits provide the parallelism
    const long blocks2 = (long)B * ((N + 2 * kBlock - 1) / (2 * kBlock));
    return blocks2 >= 1024;
}

// ---- softmin ---------------------------------------------------------------------------------------

template <int D, int P, bool DIRECT, bool BWD, typename T>
void launch_softmin_r(const SoftminParams<T>& prm, const Ranges& rg, int n_ranges, int B, int N, int M,
                      const Scratch& sc, hipStream_t st) {
    if (use_two_rows(B, N, n_ranges, sc)) {
        if constexpr (BWD) launch_mapreduce<SoftminBwdOp<D, P, DIRECT, 2, T>>(prm, rg, n_ranges, B, N, M, sc.ws, sc.bytes, sc.allow_split, st, sc.cb);
        else launch_mapreduce<SoftminFwdOp<D, P, DIRECT, 2, T>>(prm, rg, n_ranges, B, N, M, sc.ws, sc.bytes, sc.allow_split, st, sc.cb);
    } else {
        if constexpr (BWD) launch_mapreduce<SoftminBwdOp<D, P, DIRECT, 1, T>>(prm, rg, n_ranges, B, N, M, sc.ws, sc.bytes, sc.allow_split, st, sc.cb);
        else launch_mapreduce<SoftminFwdOp<D, P, DIRECT, 1, T>>(prm, rg, n_ranges, B, N, M, sc.ws, sc.bytes, sc.allow_split, st, sc.cb);
    }
}

// p = 2 forward on the matrix cores (glhip_softmin_mfma.h); same partial format / merge kernel as the VALU op.
// 2 row tiles per wavefront (128 rows per workgroup): 84-126 VGPRs -> 4-5 waves/SIMD; measured equal to 4 tiles at
// N=M=1e6 and 11-16 % faster on mid-size, batched and block-sparse problems.
constexpr int kFwdRT = 2;
constexpr long kFwdSlots = 256 * 3;   // resident 8-wave workgroups of the forward / gaussian x32 kernels (<= 84 VGPRs, 32 KiB LDS)
// p = 2 forward on the matrix cores; same partial format / merge kernel as the VALU op.
//   KIND 0: fp32 MFMA (glhip_softmin_mfma.h), 4 waves.   KIND 1: bf16x3 on 16x16x32 MFMAs (glhip_softmin_xdl.h).
//   KIND 2: bf16x3 on 32x32x16 MFMAs, transposed blocks (glhip_softmin_x32.h) — the default.
enum { FWD_F32 = 0, FWD_XDL16 = 1, FWD_X32 = 2 };

template <int D, typename T, int KIND, int NW, bool SPARSE, int L = XL_BF16X3>
void launch_fwd_kernel(dim3 grid, hipStream_t st, const SoftminParams<T>& prm, const Ranges& rg, int N, int M, const SplitInfo& sp) {
    if constexpr (KIND == FWD_X32) hipLaunchKernelGGL((softmin_fwd_x32_kernel<D, T, SPARSE, 1, NW, false, L>), grid, dim3(NW * 64), 0, st, prm, rg, N, M, sp, PackedCols{nullptr, 0});
    else if constexpr (KIND == FWD_XDL16) hipLaunchKernelGGL((softmin_fwd_xdl_kernel<D, T, SPARSE, kFwdRT, NW>), grid, dim3(NW * 64), 0, st, prm, rg, N, M, sp);
    else hipLaunchKernelGGL((softmin_fwd_mfma_kernel<D, T, SPARSE, kFwdRT>), grid, dim3(kBlock), 0, st, prm, rg, N, M, sp);
}

template <int D, typename T, int KIND, int NW, int L = XL_BF16X3>
void launch_softmin_mfma_nw(const SoftminParams<T>& prm, const Ranges& rg, int n_ranges, int B, int N, int M,
                            const Scratch& sc, hipStream_t st) {
    static_assert(L == XL_BF16X3 || KIND == FWD_X32, "the f16 x 2 layout exists on the 32x32x16 kernel only");
    constexpr int NR = X32Layout<L>::NR;
    using MergeOp = SoftminFwdOp<D, 2, false, 1, T>;   // the forward merge does not use the row-pass centre
    constexpr int kRowsPerBlock = NW * 32;             // 16 * kFwdRT = 32 rows per wavefront in all three kernels
    static_assert(kFwdRT == 2, "row tiling of the forward kernels");
    unsigned chunk_grid = 0;   // block-sparse: one workgroup per row chunk of kRowsPerBlock rows (build_row_chunks_kernel)
    // 4 wavefronts on the 32x32x16 kernel, f16 x 2 layout: chunks of whole groups of 4 row tiles, leftover row tiles carried (glhip_softmin_x32.h)
    // (profiles/r06_carried_tiles_ab.txt)
    const int share = (KIND == FWD_X32 && NW == 4 && L == XL_F16X2 && n_ranges > 0) ? 1 : 0;
    const Ranges rgc = n_ranges > 0 ? with_row_chunks(rg, n_ranges, N, kRowsPerBlock, sc.cb, st, chunk_grid, share) : rg;
    // the number of column splits is still derived from the number of row BLOCKS: deriving it from the (larger) chunk count
    // gives fewer, longer-lived workgroups and measured 3 % slower on uniform clusters (multiscale at 1e6: 258 vs 250 ms)
    const long row_blocks = n_ranges > 0 ? (long)n_ranges : (long)B * ((N + kRowsPerBlock - 1) / kRowsPerBlock);
    const long per_split = (long)B * N * 2 * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits(row_blocks, M, n_ranges, fit) : 1;
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = (long)B * N * 2;
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    // block-sparse: small row clusters come with short column intervals (the reference's cluster_scale rule makes ~2000 clusters
    // whatever N is) — gather them into full tiles; clusters of hundreds of points already fill theirs
    sp.gather = (n_ranges > 0 && N / n_ranges < 128) ? 1 : 0;
    sp.share = (share && rgc.chunks) ? 1 : 0;
    // 2-wavefront workgroups (small clusters): 12 of them are resident per CU, three times the 4-wavefront case choose_splits is
    // tuned for — at least 6 splits (two-scale loss at N = 3e4 / 5e4 / 1e5: 2.11 / 3.01 / 5.30 -> 2.08 / 2.96 / 5.21 ms, three runs each)
    if (NW == 2 && n_ranges > 0 && sc.allow_split && fit >= 6 && sp.n_splits > 1 && sp.n_splits < 6) sp.n_splits = 6;

    // Dense launches of the x32 kernel with enough work to pay for one more (tiny) launch split the columns into
    // bf16x3 MFMA records ONCE, in workspace behind the split partials, instead of once per workgroup.
    PackedCols pk{nullptr, (long)((M + 31) / 32) * (32 * NR)};
    const size_t packed_bytes = (size_t)B * (size_t)pk.stride * sizeof(uint4);   // either layout fits
    auto plan_pre = [&](int ns) {
        const size_t part_bytes = (((size_t)(ns > 1 ? ns : 0) * per_split) + 255) & ~(size_t)255;
        if (KIND != FWD_X32 || !sc.prepack((double)B * N * M)) return false;
        if (sc.bytes < part_bytes + packed_bytes) return false;
        pk.rec = reinterpret_cast<uint4*>(static_cast<char*>(sc.ws) + part_bytes);
        return true;
    };
    auto pack = [&]() {
        if (n_ranges > 0) hipLaunchKernelGGL((pack_columns_kernel<D, T, false, L>), dim3((M + kBlock - 1) / kBlock, B, 1), dim3(kBlock), 0, st, prm, N, M, pk);
        else hipLaunchKernelGGL((pack_columns_kernel<D, T, true, L>), dim3((M + 31 + kBlock) / kBlock, B, 1), dim3(kBlock), 0, st, prm, N, M, pk);
    };

    if constexpr (NW == 2) {    // block-sparse launches on row blocks of up to 64 points only (launch_softmin_mfma)
        if (plan_pre(sp.n_splits)) {
            pack();
            hipLaunchKernelGGL((softmin_fwd_x32_kernel<D, T, true, 1, NW, true, L>), dim3(chunk_grid, 1, sp.n_splits), dim3(NW * 64), 0, st, prm, rgc, N, M, sp, pk);
        } else {
            hipLaunchKernelGGL((softmin_fwd_x32_kernel<D, T, true, 1, NW, false, L>), dim3(chunk_grid, 1, sp.n_splits), dim3(NW * 64), 0, st, prm, rgc, N, M, sp, PackedCols{nullptr, 0});
        }
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<MergeOp, true>), dim3(n_ranges, 1, 1), dim3(kBlock), 0, st, prm, rg, N, sp);
    } else {
    if (KIND != FWD_F32 && n_ranges == 0 && sc.allow_split && fit >= 8 && M >= 65536) {
        // large dense problem: exactly 8 column splits, one per XCD (see workgroup_coords)
        const int gx = (N + kRowsPerBlock - 1) / kRowsPerBlock;
        const int nx = (KIND == FWD_X32 && sc.prepack((double)B * N * M)) ? xcd_splits_prepacked((long)gx * B, M, kFwdSlots, fit, NR * 16.0)
                                                                          : xcd_splits((long)gx * B, M, kFwdSlots, fit);
        const long total = (long)gx * B * nx;
        if (total < (1L << 31)) {
            sp.n_splits = nx;
            sp.xcd_grid_x = gx;
            sp.xcd_blocks = gx * B;
            if (plan_pre(nx)) {
                pack();
                hipLaunchKernelGGL((softmin_fwd_x32_kernel<D, T, false, 1, NW, true, L>), dim3((unsigned)total, 1, 1), dim3(NW * 64), 0, st, prm, rg, N, M, sp, pk);
            } else {
                launch_fwd_kernel<D, T, KIND, NW, false, L>(dim3((unsigned)total, 1, 1), st, prm, rg, N, M, sp);
            }
            hipLaunchKernelGGL((merge_kernel<MergeOp, false>), dim3((N + kBlock - 1) / kBlock, B, 1), dim3(kBlock), 0, st, prm, rg, N, sp);
            return;
        }
    }
    if (n_ranges > 0) {
        if (plan_pre(sp.n_splits)) {
            pack();
            hipLaunchKernelGGL((softmin_fwd_x32_kernel<D, T, true, 1, NW, true, L>), dim3(chunk_grid, 1, sp.n_splits), dim3(NW * 64), 0, st, prm, rgc, N, M, sp, pk);
        } else {
            launch_fwd_kernel<D, T, KIND, NW, true, L>(dim3(chunk_grid, 1, sp.n_splits), st, prm, rgc, N, M, sp);
        }
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<MergeOp, true>), dim3(n_ranges, 1, 1), dim3(kBlock), 0, st, prm, rg, N, sp);
    } else {
        const int gx = (N + kRowsPerBlock - 1) / kRowsPerBlock;
        if (plan_pre(sp.n_splits)) {
            pack();
            hipLaunchKernelGGL((softmin_fwd_x32_kernel<D, T, false, 1, NW, true, L>), dim3(gx, B, sp.n_splits), dim3(NW * 64), 0, st, prm, rg, N, M, sp, pk);
        } else {
            launch_fwd_kernel<D, T, KIND, NW, false, L>(dim3(gx, B, sp.n_splits), st, prm, rg, N, M, sp);
        }
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<MergeOp, false>), dim3((N + kBlock - 1) / kBlock, B, 1), dim3(kBlock), 0, st, prm, rg, N, sp);
    }
    }
}

template <int D, typename T, int KIND>
void launch_softmin_mfma(const SoftminParams<T>& prm, const Ranges& rg, int n_ranges, int B, int N, int M,
                         const Scratch& sc, hipStream_t st) {
    // Workgroup height of the bf16x3 kernels: 8 wavefronts (256 rows per pass) for launches big enough to run with
    // pre-packed columns — dense (1-4 % faster than 4 there, measured from B x N = 256 x 4096 to 1 x 1e6) and
    // block-sparse with row blocks of a few hundred points (multiscale at 1e6: 0.27 vs 0.30 s); 4 wavefronts when
    // every workgroup packs its own tiles or the row blocks are small, where more, smaller workgroups win.
    static const int forced_nw = getenv("GLHIP_FWD_NW") ? atoi(getenv("GLHIP_FWD_NW")) : 0;   // tuning knob (4 or 8)
    if constexpr (KIND == FWD_X32) {
        if (sc.h2) {      // GLHIP_FLAG_F16X2: the same kernel on the f16 x 2 layout (one MFMA per block); same workgroup shapes
            // f16 x 2 (16 KB tiles): block-sparse launches run 4-wavefront workgroups whatever the cluster size — at N = 1e6 (clusters of
            // ~455 rows) 236 ms per two-scale loss against 241 with 8 wavefronts and 297 with 2 (round 6; with 32-KB bf16 x 3 tiles
            // 8 wavefronts won: the rule below); 2 x 2 row tiles per wavefront and 2 .. 12 column splits made no difference
            if (n_ranges > 0 && ((sc.small_rows && !forced_nw) || forced_nw == 2)) launch_softmin_mfma_nw<D, T, FWD_X32, 2, XL_F16X2>(prm, rg, n_ranges, B, N, M, sc, st);
            else if (forced_nw ? forced_nw == 8 : (n_ranges == 0 && (double)B * N * M >= 5e8 && (long)B * N >= 32768))
                launch_softmin_mfma_nw<D, T, FWD_X32, 8, XL_F16X2>(prm, rg, n_ranges, B, N, M, sc, st);
            else launch_softmin_mfma_nw<D, T, FWD_X32, 4, XL_F16X2>(prm, rg, n_ranges, B, N, M, sc, st);
            return;
        }
    }
    if constexpr (KIND == FWD_F32)
        launch_softmin_mfma_nw<D, T, FWD_F32, 4>(prm, rg, n_ranges, B, N, M, sc, st);
    else if (KIND == FWD_X32 && n_ranges > 0 && sc.small_rows && !forced_nw)
        // the caller says the pairs sit in row blocks of up to 64 points (GLHIP_FLAG_SMALL_ROW_BLOCKS): 2 wavefronts x 256-column
        // tiles — in a 4-wavefront workgroup half the wavefronts would own no row and only stage and wait.  Not inferred from the
        // MEAN block (N / n_ranges): clusters of a cloud sampled on a surface average 47 points at N = 1e5 while most pairs
        // belong to blocks of hundreds, which 64-row workgroups cut into twice the chunks (measured: 12.2 -> 18.1 ms per loss)
        launch_softmin_mfma_nw<D, T, FWD_X32, 2>(prm, rg, n_ranges, B, N, M, sc, st);
    else if (forced_nw ? forced_nw == 8
                       : ((double)B * N * M >= 5e8 && (n_ranges == 0 ? (long)B * N >= 32768 : N / n_ranges >= 192)))
        launch_softmin_mfma_nw<D, T, KIND == FWD_F32 ? FWD_XDL16 : KIND, 8>(prm, rg, n_ranges, B, N, M, sc, st);
    else
        launch_softmin_mfma_nw<D, T, KIND == FWD_F32 ? FWD_XDL16 : KIND, 4>(prm, rg, n_ranges, B, N, M, sc, st);
}

// weighted-sum matrix-core kernels; MergeOp is the VALU operator with the same partial format.
//   x32 = true: glhip_wsum_x32.h (32x32x16 MFMAs, pre-packed columns when the launch is big enough) — the default;
//   x32 = false: glhip_wsum_mfma.h (16x16x32 MFMAs, GLHIP_FLAG_XDL16).
// The 32x32x16 form pays for the one-component reduction (gaussian product: 1 exp2 + 1 fma per pair, 89.8 vs 92.4 ms
// at 1e6).  With D + 1 accumulators per row it needs 64 accumulator registers per lane and its bare loop measures
// 22.8 cycles per 64 pairs (tools/ubench/overlap.hip) — what the 16x16x32 kernel already delivers end to end
// (23.3); the shipped x32 gradient kernels were slower (188 vs 148 ms), so the gradients stay on glhip_wsum_mfma.h.
template <int MODE> constexpr bool wsum_uses_x32() { return MODE == WS_GAUSS_FWD; }

template <int MODE, int D, typename T, bool SPARSE>
void launch_wsum_kernel(bool x32, bool pre, dim3 grid, hipStream_t st, const WsumParams<T>& prm, const Ranges& rg, int N, int M,
                        const SplitInfo& sp, const PackedCols& pk, const PackedQ& pq) {
    if constexpr (wsum_uses_x32<MODE>()) {
        if (x32 && pre) { hipLaunchKernelGGL((wsum_x32_kernel<MODE, D, T, SPARSE, true>), grid, dim3(kWsumNW * 64), 0, st, prm, rg, N, M, sp, pk, pq); return; }
        if (x32) { hipLaunchKernelGGL((wsum_x32_kernel<MODE, D, T, SPARSE, false>), grid, dim3(kWsumNW * 64), 0, st, prm, rg, N, M, sp, pk, pq); return; }
    }
    // block-sparse soft-min gradient / value + gradient: 2 row tiles x 8 wavefronts (119 VGPRs, 4 wavefronts per SIMD; the last
    // wavefronts of a partial chunk own no rows and skip the arithmetic) — 42.2 -> 40.4 ms at N = 1e6; dense launches are faster on
    // 4 x 4 (149 vs 160 ms), the gaussian modes indifferent (164 vs 163)
    if constexpr (SPARSE && MODE == WS_SOFTMIN_BWD) hipLaunchKernelGGL((wsum_mfma_kernel<MODE, D, T, SPARSE, 2, 8>), grid, dim3(512), 0, st, prm, rg, N, M, sp);
    else hipLaunchKernelGGL((wsum_mfma_kernel<MODE, D, T, SPARSE>), grid, dim3(kBlock), 0, st, prm, rg, N, M, sp);
}

template <int MODE, int D, typename T, class MergeOp>
void launch_wsum(const WsumParams<T>& prm, const typename MergeOp::Params& mprm, const Ranges& rg, int n_ranges, int B,
                 int N, int M, const Scratch& sc, bool x32, hipStream_t st) {
    static_assert(kMfmaRowsPerBlock == kBlock * MergeOp::kRows, "merge kernel and MFMA kernel must tile rows alike");
    static_assert(WsumShape<MODE, D>::kPart == MergeOp::kPartial, "partial formats differ");
    constexpr int NQ = WsumShape<MODE, D>::kNQ;
    unsigned chunk_grid = 0;
    const Ranges rgc = n_ranges > 0 ? with_row_chunks(rg, n_ranges, N, kMfmaRowsPerBlock, sc.cb, st, chunk_grid) : rg;
    const long row_blocks = n_ranges > 0 ? (long)n_ranges : (long)B * ((N + kMfmaRowsPerBlock - 1) / kMfmaRowsPerBlock);
    const long per_split = (long)B * N * MergeOp::kPartial * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits(row_blocks, M, n_ranges, fit) : 1;
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = (long)B * N * MergeOp::kPartial;
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;

    // pre-packed column records + q vectors behind the split partials (see launch_softmin_mfma_nw)
    PackedCols pk{nullptr, (long)((M + 31) / 32) * 128};
    PackedQ pq{nullptr, (long)B * M};
    const size_t rec_bytes = (size_t)B * (size_t)pk.stride * sizeof(uint4);
    auto plan_pre = [&](int ns) {
        const size_t part_bytes = (((size_t)(ns > 1 ? ns : 0) * per_split) + 255) & ~(size_t)255;
        if (!wsum_uses_x32<MODE>() || !x32 || !sc.prepack((double)B * N * M)) return false;
        if (sc.bytes < part_bytes + rec_bytes + (size_t)NQ * B * M * sizeof(float)) return false;
        pk.rec = reinterpret_cast<uint4*>(static_cast<char*>(sc.ws) + part_bytes);
        pq.q = reinterpret_cast<float*>(static_cast<char*>(sc.ws) + part_bytes + rec_bytes);
        if constexpr (wsum_uses_x32<MODE>()) {
            if (n_ranges > 0) hipLaunchKernelGGL((wsum_pack_kernel<MODE, D, T, false>), dim3((M + kBlock - 1) / kBlock, B, 1), dim3(kBlock), 0, st, prm, N, M, pk, pq);
            else hipLaunchKernelGGL((wsum_pack_kernel<MODE, D, T, true>), dim3((M + 31 + kBlock) / kBlock, B, 1), dim3(kBlock), 0, st, prm, N, M, pk, pq);
        }
        return true;
    };

    if (n_ranges == 0 && sc.allow_split && fit >= 8 && M >= 65536) {   // one column split per XCD (workgroup_coords)
        const int gx = (N + kMfmaRowsPerBlock - 1) / kMfmaRowsPerBlock;
        const int nx = !(wsum_uses_x32<MODE>() && x32) ? 8
                       : sc.prepack((double)B * N * M) ? xcd_splits_prepacked((long)gx * B, M, kFwdSlots, fit)
                                                       : xcd_splits((long)gx * B, M, kFwdSlots, fit);
        const long total = (long)gx * B * nx;
        if (total < (1L << 31)) {
            sp.n_splits = nx;
            sp.xcd_grid_x = gx;
            sp.xcd_blocks = gx * B;
            const bool pre = plan_pre(nx);
            launch_wsum_kernel<MODE, D, T, false>(x32, pre, dim3((unsigned)total, 1, 1), st, prm, rg, N, M, sp, pk, pq);
            hipLaunchKernelGGL((merge_kernel<MergeOp, false>), dim3(gx, B, 1), dim3(kBlock), 0, st, mprm, rg, N, sp);
            return;
        }
    }
    const bool pre = plan_pre(sp.n_splits);
    if (n_ranges > 0) {
        launch_wsum_kernel<MODE, D, T, true>(x32, pre, dim3(chunk_grid, 1, sp.n_splits), st, prm, rgc, N, M, sp, pk, pq);
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<MergeOp, true>), dim3(n_ranges, 1, 1), dim3(kBlock), 0, st, mprm, rg, N, sp);
    } else {
        const int gx = (N + kMfmaRowsPerBlock - 1) / kMfmaRowsPerBlock;
        launch_wsum_kernel<MODE, D, T, false>(x32, pre, dim3(gx, B, sp.n_splits), st, prm, rg, N, M, sp, pk, pq);
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<MergeOp, false>), dim3(gx, B, 1), dim3(kBlock), 0, st, mprm, rg, N, sp);
    }
}

template <int D, typename T>
void launch_softmin_bwd_mfma(const SoftminParams<T>& prm, const Ranges& rg, int n_ranges, int B, int N, int M,
                             const Scratch& sc, bool x32, hipStream_t st) {
    WsumParams<T> w;
    w.x = prm.x; w.y = prm.y; w.s = prm.h; w.fwd = prm.fwd; w.g = prm.g; w.out = prm.out; w.gx = prm.gx;
    w.s2 = prm.s2; w.out_scale = prm.out_scale; w.gscale = 1.f; w.tscale = prm.shift2;
    launch_wsum<WS_SOFTMIN_BWD, D, T, SoftminBwdOp<D, 2, false, 1, T>>(w, prm, rg, n_ranges, B, N, M, sc, x32, st);
}

// p = 1 soft-min / laplacian / energy with the squared distance on the matrix cores (glhip_dist_x32.h): block-sparse launches whose
// row blocks are spatially compact, on the caller's word (GLHIP_FLAG_MFMA_DIST).  Row chunks, column splits and the merge as above.
constexpr int kDistNW = 8;
inline float dist_guard() {   // GLHIP_DIST_GUARD: test knob (1e30 = every pair on explicit differences, 0 = none)
    static const float g = getenv("GLHIP_DIST_GUARD") ? (float)atof(getenv("GLHIP_DIST_GUARD")) : 1.0f / 256.0f;
    return g;
}

template <int MODE, int D, typename T, class MergeOp, bool FAMILY = false>
void launch_dist(const DistParams<T>& prm, const typename MergeOp::Params& mprm, const Ranges& rg, int n_ranges, int N, int M,
                 const Scratch& sc, hipStream_t st) {
    unsigned chunk_grid = 0;
    const Ranges rgc = with_row_chunks(rg, n_ranges, N, kDistNW * 32, sc.cb, st, chunk_grid);
    const long per_split = (long)N * MergeOp::kPartial * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits(n_ranges, M, n_ranges, fit) : 1;
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = (long)N * MergeOp::kPartial;
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    hipLaunchKernelGGL((dist_x32_kernel<MODE, D, T, kDistNW, FAMILY>), dim3(chunk_grid, 1, sp.n_splits), dim3(kDistNW * 64), 0, st, prm, rgc, N, M, sp);
    if (sp.n_splits > 1)
        hipLaunchKernelGGL((merge_kernel<MergeOp, true>), dim3(n_ranges, 1, 1), dim3(kBlock), 0, st, mprm, rg, N, sp);
}

// laplacian / energy product + gradient (GM = DG_FWDGRAD) or gradient (DG_BWD) on matrix-core distances (glhip_dist_grad_x32.h);
// MergeOp = ConvOp<KIND, D, 1, T, 2 | 1>: same partial formats as the direct-difference operators
template <int KIND, int GM, int D, typename T>
void launch_dist_grad(const ConvParams<T>& prm, const Ranges& rg, int n_ranges, int N, int M, const Scratch& sc, hipStream_t st) {
    using MergeOp = ConvOp<KIND, D, 1, T, GM == DG_FWDGRAD ? 2 : 1>;
    DistGradParams<T> gp;
    gp.d = DistParams<T>{prm.x, prm.y, prm.v, nullptr, nullptr, prm.out, prm.t, prm.clamp2, 1.f, 0.f, 1.f, 0.f, dist_guard()};
    gp.g = prm.g;
    gp.gx = prm.gx;
    gp.gscale = prm.gscale;
    unsigned chunk_grid = 0;
    const Ranges rgc = with_row_chunks(rg, n_ranges, N, kDistNW * 32, sc.cb, st, chunk_grid);
    const long per_split = (long)N * MergeOp::kPartial * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits(n_ranges, M, n_ranges, fit) : 1;
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = (long)N * MergeOp::kPartial;
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    hipLaunchKernelGGL((dist_grad_x32_kernel<KIND, GM, D, T, kDistNW>), dim3(chunk_grid, 1, sp.n_splits), dim3(kDistNW * 64), 0, st, gp, rgc, N, M, sp);
    if (sp.n_splits > 1)
        hipLaunchKernelGGL((merge_kernel<MergeOp, true>), dim3(n_ranges, 1, 1), dim3(kBlock), 0, st, prm, rg, N, sp);
}

template <int KIND, int GM, typename T>
void launch_dist_grad_d(const ConvParams<T>& prm, const Ranges& rg, int n_ranges, int N, int M, int D, const Scratch& sc, hipStream_t st) {
    if (D == 1) launch_dist_grad<KIND, GM, 1, T>(prm, rg, n_ranges, N, M, sc, st);
    else if (D == 2) launch_dist_grad<KIND, GM, 2, T>(prm, rg, n_ranges, N, M, sc, st);
    else launch_dist_grad<KIND, GM, 3, T>(prm, rg, n_ranges, N, M, sc, st);
}

// ---- 4 <= D <= 16 on the matrix cores (glhip_softmin_xd.h): soft-min forward / fused half-step (MODE XD_SOFTMIN) and gaussian
// product (XD_GAUSS).  Column splits, XCD-aware 1-D grid for big dense launches, row chunks for block-sparse ones and the merge
// kernels are those of the D <= 3 kernels; there is no pre-packed column copy.
constexpr int kXdMaxD = 16;
constexpr long kXdSlots = 256 * 2;    // resident 8-wave workgroups (<= 48 KiB of LDS, <= 128 VGPRs)

template <int MODE, int D, typename T, class MergeOp, int RT, int NW, int L>
void launch_xd_cfg(const SoftminParams<T>& prm, const typename MergeOp::Params& mprm, const Ranges& rg, int n_ranges, int B, int N,
                   int M, const Scratch& sc, hipStream_t st) {
    using S = XdShape<D, L>;
    constexpr int kPart = MODE == XD_SOFTMIN ? 2 : 1;
    static_assert(MergeOp::kPartial == kPart, "partial formats differ");
    static_assert(MergeOp::kRows == 1, "the merge launch below tiles rows in blocks of kBlock");
    constexpr int kRows = RT * NW * 32;
    unsigned chunk_grid = 0;
    const Ranges rgc = n_ranges > 0 ? with_row_chunks(rg, n_ranges, N, kRows, sc.cb, st, chunk_grid) : rg;
    const long row_blocks = n_ranges > 0 ? (long)n_ranges : (long)B * ((N + kRows - 1) / kRows);
    const long per_split = (long)B * N * kPart * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits(row_blocks, M, n_ranges, fit) : 1;
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = (long)B * N * kPart;
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    sp.gather = (n_ranges > 0 && N / n_ranges < 128) ? 1 : 0;   // small clusters: gathered tiles, as in launch_softmin_mfma_nw
    const int gx = (N + kRows - 1) / kRows;
    const dim3 merge_grid((N + kBlock - 1) / kBlock, B, 1);
    const XdPacked none{nullptr, 0};
    // pre-packed columns (glhip_softmin_xd.h): the records of all columns once, in workspace behind the split partials
    XdPacked pk{nullptr, (long)((M + 31) / 32) * S::kGroupRecs};
    const size_t packed_bytes = (size_t)B * (size_t)pk.stride * sizeof(uint4);
    const long fit_pre = (sc.ws && sc.bytes > packed_bytes + 256) ? (long)((sc.bytes - packed_bytes - 256) / per_split) : 0;
    const bool pre = NW == 8 && sc.prepack((double)B * N * M) && fit_pre >= 8;      // (A/B knob GLHIP_XD_PRE of rounds 4-5: pre-packed columns won)
    if constexpr (NW == 8) {
        // Packed columns that fit every XCD's L2 (<= 3.5 MB: M = 1e5 at 32 bytes per column) need no XCD-aware placement: any number
        // of splits, on the plain 3-D grid, chosen to fill the chip's rounds of resident workgroups (free_splits).  BASELINE config 2,
        // online N = M = 1e5: 13 splits instead of 32: 0.937 -> 0.896 ms per soft-min, 35.3 -> 34.2 ms per loss; N = 7e4: 0.489 -> 0.439 ms.
        // From M = 8192 (below 65536 the launch used to pack its columns per workgroup, on up to 32 splits: raw soft-mins at N = M = 5e4,
        // D = 4 / 5 / 8: 0.318 / 0.402 / 0.396 -> 0.280 / 0.303 / 0.300 ms).
        if (n_ranges == 0 && sc.allow_split && pre && M >= 8192 && (double)M * S::NBP * 16.0 <= 3.5e6) {
            double eff_free = 0.0;
            const int nf = free_splits((long)gx * B, M, kXdSlots, fit_pre, &eff_free);
            double eff_x = 0.0;
            if (M >= 65536) {
                const int nx8 = xcd_splits_prepacked((long)gx * B, M, kXdSlots, fit_pre, S::NBP * 16.0);
                eff_x = ((double)gx * B * nx8 / (double)kXdSlots) / (double)(((long)gx * B * nx8 + kXdSlots - 1) / kXdSlots);
            }
            if (nf >= 1 && eff_free > eff_x + 0.02) {
                sp.n_splits = nf;
                const size_t part_bytes = (((size_t)(nf > 1 ? nf : 0) * per_split) + 255) & ~(size_t)255;
                pk.rec = reinterpret_cast<uint4*>(static_cast<char*>(sc.ws) + part_bytes);
                hipLaunchKernelGGL((xd_pack_kernel<MODE, D, T, L>), dim3((M + 31 + kBlock) / kBlock, B, 1), dim3(kBlock), 0, st, prm, N, M, pk);
                hipLaunchKernelGGL((xd_fwd_kernel<MODE, D, T, false, RT, NW, true, L>), dim3(gx, B, nf), dim3(NW * 64), 0, st, prm, rg, N, M, sp, pk);
                if (nf > 1) hipLaunchKernelGGL((merge_kernel<MergeOp, false>), merge_grid, dim3(kBlock), 0, st, mprm, rg, N, sp);
                return;
            }
        }
    }
    if (n_ranges == 0 && sc.allow_split && fit >= 8 && M >= 65536) {   // one column split per XCD at a time (workgroup_coords)
        const int nx = pre ? xcd_splits_prepacked((long)gx * B, M, kXdSlots, fit_pre, S::NBP * 16.0) : xcd_splits((long)gx * B, M, kXdSlots, fit);
        const long total = (long)gx * B * nx;
        if (total < (1L << 31)) {
            sp.n_splits = nx;
            sp.xcd_grid_x = gx;
            sp.xcd_blocks = gx * B;
            const size_t part_bytes = (((size_t)nx * per_split) + 255) & ~(size_t)255;
            if (pre) {
                pk.rec = reinterpret_cast<uint4*>(static_cast<char*>(sc.ws) + part_bytes);
                hipLaunchKernelGGL((xd_pack_kernel<MODE, D, T, L>), dim3((M + 31 + kBlock) / kBlock, B, 1), dim3(kBlock), 0, st, prm, N, M, pk);
                if constexpr (NW == 8)
                    hipLaunchKernelGGL((xd_fwd_kernel<MODE, D, T, false, RT, NW, true, L>), dim3((unsigned)total, 1, 1), dim3(NW * 64), 0, st, prm, rg, N, M, sp, pk);
            } else {
                hipLaunchKernelGGL((xd_fwd_kernel<MODE, D, T, false, RT, NW, false, L>), dim3((unsigned)total, 1, 1), dim3(NW * 64), 0, st, prm, rg, N, M, sp, none);
            }
            hipLaunchKernelGGL((merge_kernel<MergeOp, false>), merge_grid, dim3(kBlock), 0, st, mprm, rg, N, sp);
            return;
        }
    }
    if (n_ranges > 0) {
        hipLaunchKernelGGL((xd_fwd_kernel<MODE, D, T, true, RT, NW, false, L>), dim3(chunk_grid, 1, sp.n_splits), dim3(NW * 64), 0, st, prm, rgc, N, M, sp, none);
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<MergeOp, true>), dim3(n_ranges, 1, 1), dim3(kBlock), 0, st, mprm, rg, N, sp);
    } else {
        hipLaunchKernelGGL((xd_fwd_kernel<MODE, D, T, false, RT, NW, false, L>), dim3(gx, B, sp.n_splits), dim3(NW * 64), 0, st, prm, rg, N, M, sp, none);
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<MergeOp, false>), merge_grid, dim3(kBlock), 0, st, mprm, rg, N, sp);
    }
}

template <int MODE, int D, typename T, class MergeOp, int L>
void launch_xd_l(const SoftminParams<T>& prm, const typename MergeOp::Params& mprm, const Ranges& rg, int n_ranges, int B, int N, int M,
                 const Scratch& sc, hipStream_t st) {
    // big launches: 8 wavefronts x 2 row tiles (the two tiles share the LDS reads of a column group; 512 rows share the bf16
    // pieces of a column when they are packed on the fly) while the x-side operands of two tiles fit 128 VGPRs — 4 waves per
    // SIMD: up to 5 chained MFMAs (D <= 12) on dense launches, 4 (D <= 9) on block-sparse ones, 1 tile beyond; small launches:
    // 4 wavefronts x 1 tile, more workgroups
    constexpr int NM = XdShape<D, L>::NM;
    const bool big = (double)B * N * M >= 5e8 && (n_ranges == 0 ? (long)B * N >= 32768 : N / n_ranges >= 192);
    if (big) {      // (constexpr where the shape decides: the configurations a dimension never takes are not compiled)
        if constexpr (NM <= 4) {
            launch_xd_cfg<MODE, D, T, MergeOp, 2, 8, L>(prm, mprm, rg, n_ranges, B, N, M, sc, st);
        } else if constexpr (NM == 5) {
            if (n_ranges == 0) launch_xd_cfg<MODE, D, T, MergeOp, 2, 8, L>(prm, mprm, rg, n_ranges, B, N, M, sc, st);
            else launch_xd_cfg<MODE, D, T, MergeOp, 1, 8, L>(prm, mprm, rg, n_ranges, B, N, M, sc, st);
        } else {
            launch_xd_cfg<MODE, D, T, MergeOp, 1, 8, L>(prm, mprm, rg, n_ranges, B, N, M, sc, st);
        }
    } else {
        launch_xd_cfg<MODE, D, T, MergeOp, 1, 4, L>(prm, mprm, rg, n_ranges, B, N, M, sc, st);
    }
}

// ... in the K layout the call asks for: bf16 x 3 (default) or f16 x 2 (GLHIP_FLAG_F16X2: the caller vouches for the range)
template <int MODE, int D, typename T, class MergeOp>
void launch_xd(const SoftminParams<T>& prm, const typename MergeOp::Params& mprm, const Ranges& rg, int n_ranges, int B, int N, int M,
               const Scratch& sc, hipStream_t st) {
    if (sc.h2) launch_xd_l<MODE, D, T, MergeOp, XL_F16X2>(prm, mprm, rg, n_ranges, B, N, M, sc, st);
    else launch_xd_l<MODE, D, T, MergeOp, XL_BF16X3>(prm, mprm, rg, n_ranges, B, N, M, sc, st);
}

// distance reductions for 4 <= D <= 16, dense launches (glhip_dist_xd.h): soft-min p = 1 / fused half-step, laplacian and energy products
// 256-row workgroups of the distance kernels on a few thousand points are a handful (N = 1000: 4 per problem): launches with fewer than
// 512 workgroups split their columns down to 128 per split, up to 32 splits and what the workspace holds
// (online p = 1 losses at N = 1000 / 2000: 0.53 / 0.49 -> 0.35 / 0.38 ms)
static inline int dist_small_launch_splits(int n_splits, long row_blocks, int M, long fit, bool allow_split) {
    constexpr int min_cols = 128;
    if (!allow_split || min_cols <= 0 || row_blocks * n_splits >= 512) return n_splits;
    long want = (512 + row_blocks - 1) / row_blocks;
    const long by_cols = M / min_cols;
    want = want < by_cols ? want : by_cols;
    want = want < 32 ? want : 32;
    want = want < fit ? want : fit;
    return want > n_splits ? (int)want : n_splits;
}

template <int MODE, int D, typename T, class MergeOp>
void launch_dist_xd(const DistParams<T>& prm, const typename MergeOp::Params& mprm, int B, int N, int M, const Scratch& sc, hipStream_t st) {
    constexpr int NW = 8, kRows = NW * 32;
    constexpr int kPart = MODE == DM_SOFTMIN_P1 ? 2 : 1;
    static_assert(MergeOp::kPartial == kPart && MergeOp::kRows == 1, "partial formats differ");
    const int gx = (N + kRows - 1) / kRows;
    const long per_split = (long)B * N * kPart * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits((long)gx * B, M, 0, fit) : 1;
    sp.n_splits = dist_small_launch_splits(sp.n_splits, (long)gx * B, M, fit, sc.allow_split);
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = (long)B * N * kPart;
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    const Ranges none{nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL((dist_xd_kernel<MODE, D, T, NW>), dim3(gx, B, sp.n_splits), dim3(NW * 64), 0, st, prm, N, M, sp);
    if (sp.n_splits > 1)
        hipLaunchKernelGGL((merge_kernel<MergeOp, false>), dim3((N + kBlock - 1) / kBlock, B, 1), dim3(kBlock), 0, st, mprm, none, N, sp);
}

// ... and their gradients with respect to the row points (dist_xd_grad_kernel)
template <int MODE, int D, typename T, class MergeOp>
void launch_dist_xd_grad(const DistXdGradParams<T>& gp, const typename MergeOp::Params& mprm, int B, int N, int M, const Scratch& sc,
                         hipStream_t st) {
    constexpr int NW = 8, kRows = NW * 32;
    constexpr int kPart = MODE == DM_SOFTMIN_P1 ? D + 1 : D;
    static_assert(MergeOp::kPartial == kPart && MergeOp::kRows == 1, "partial formats differ");
    const int gx = (N + kRows - 1) / kRows;
    const long per_split = (long)B * N * kPart * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits((long)gx * B, M, 0, fit) : 1;
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = (long)B * N * kPart;
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    const Ranges none{nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL((dist_xd_grad_kernel<MODE, D, T, NW>), dim3(gx, B, sp.n_splits), dim3(NW * 64), 0, st, gp, N, M, sp);
    if (sp.n_splits > 1)
        hipLaunchKernelGGL((merge_kernel<MergeOp, false>), dim3((N + kBlock - 1) / kBlock, B, 1), dim3(kBlock), 0, st, mprm, none, N, sp);
}

// weighted-sum reductions on transposed 32 x 32 blocks (glhip_wsum_t32.h), 1 <= D <= 16; splits / grids / merges as launch_wsum
// WQ: the weighted sums on the matrix cores too (wsum_t32q_kernel: soft-min gradient, f16 x 2, one row tile per wavefront)
template <int MODE, int D, typename T, class MergeOp, int RT, int L, bool WQ = false>
void launch_wsum_t32_rt(const WsumParams<T>& prm, const typename MergeOp::Params& mprm, const Ranges& rg, int n_ranges, int B, int N,
                     int M, const Scratch& sc, hipStream_t st) {
    constexpr int kPart = (MODE == WS_GAUSS_BWD) ? D : D + 1;
    static_assert(MergeOp::kPartial == kPart, "partial formats differ");
    static_assert(kMfmaRowsPerBlock == kBlock * MergeOp::kRows, "merge kernel and main kernel must tile rows alike");
    constexpr int NW = 8 / RT;       // 256 rows per workgroup either way
    unsigned chunk_grid = 0;
    const Ranges rgc = n_ranges > 0 ? with_row_chunks(rg, n_ranges, N, kMfmaRowsPerBlock, sc.cb, st, chunk_grid) : rg;
    const long row_blocks = n_ranges > 0 ? (long)n_ranges : (long)B * ((N + kMfmaRowsPerBlock - 1) / kMfmaRowsPerBlock);
    const long per_split = (long)B * N * kPart * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits(row_blocks, M, n_ranges, fit) : 1;
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = (long)B * N * kPart;
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    const int gx = (N + kMfmaRowsPerBlock - 1) / kMfmaRowsPerBlock;
    if (n_ranges == 0 && sc.allow_split && fit >= 8 && M >= 65536) {
        const int nx = xcd_splits((long)gx * B, M, kXdSlots, fit);
        const long total = (long)gx * B * nx;
        if (total < (1L << 31)) {
            sp.n_splits = nx;
            sp.xcd_grid_x = gx;
            sp.xcd_blocks = gx * B;
            if constexpr (WQ) hipLaunchKernelGGL((wsum_t32q_kernel<D, T, false, NW>), dim3((unsigned)total, 1, 1), dim3(NW * 64), 0, st, prm, rg, N, M, sp);
            else hipLaunchKernelGGL((wsum_t32_kernel<MODE, D, T, false, RT, NW, L>), dim3((unsigned)total, 1, 1), dim3(NW * 64), 0, st, prm, rg, N, M, sp);
            hipLaunchKernelGGL((merge_kernel<MergeOp, false>), dim3(gx, B, 1), dim3(kBlock), 0, st, mprm, rg, N, sp);
            return;
        }
    }
    if (n_ranges > 0) {
        if constexpr (WQ) hipLaunchKernelGGL((wsum_t32q_kernel<D, T, true, NW>), dim3(chunk_grid, 1, sp.n_splits), dim3(NW * 64), 0, st, prm, rgc, N, M, sp);
        else hipLaunchKernelGGL((wsum_t32_kernel<MODE, D, T, true, RT, NW, L>), dim3(chunk_grid, 1, sp.n_splits), dim3(NW * 64), 0, st, prm, rgc, N, M, sp);
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<MergeOp, true>), dim3(n_ranges, 1, 1), dim3(kBlock), 0, st, mprm, rg, N, sp);
    } else {
        if constexpr (WQ) hipLaunchKernelGGL((wsum_t32q_kernel<D, T, false, NW>), dim3(gx, B, sp.n_splits), dim3(NW * 64), 0, st, prm, rg, N, M, sp);
        else hipLaunchKernelGGL((wsum_t32_kernel<MODE, D, T, false, RT, NW, L>), dim3(gx, B, sp.n_splits), dim3(NW * 64), 0, st, prm, rg, N, M, sp);
        if (sp.n_splits > 1)
            hipLaunchKernelGGL((merge_kernel<MergeOp, false>), dim3(gx, B, 1), dim3(kBlock), 0, st, mprm, rg, N, sp);
    }
}

template <int MODE, int D, typename T, class MergeOp>
void launch_wsum_t32(const WsumParams<T>& prm, const typename MergeOp::Params& mprm, const Ranges& rg, int n_ranges, int B, int N,
                     int M, const Scratch& sc, hipStream_t st) {
    // 2 row tiles per wavefront share the LDS reads of a column group (4 wavefronts x 64 rows) up to D = 8 — measured 3-16 % faster
    // than 1 tile there (profiles/r03_grad_kernels_ab.txt); beyond, the x-side operands of two tiles no longer fit 128 VGPRs
    if constexpr (MODE == WS_SOFTMIN_BWD) {
        constexpr int wq_min_d = 7;      // (measured in round 5: the matrix-core weighted sums pay from D = 7 on)
        if (sc.h2 && D >= wq_min_d) {
            launch_wsum_t32_rt<MODE, D, T, MergeOp, 1, XL_F16X2, true>(prm, mprm, rg, n_ranges, B, N, M, sc, st);
            return;
        }
    }
    if (sc.h2) launch_wsum_t32_rt<MODE, D, T, MergeOp, (D <= 8 ? 2 : 1), XL_F16X2>(prm, mprm, rg, n_ranges, B, N, M, sc, st);
    else launch_wsum_t32_rt<MODE, D, T, MergeOp, (D <= 8 ? 2 : 1), XL_BF16X3>(prm, mprm, rg, n_ranges, B, N, M, sc, st);
}

// soft-min gradient (and value + gradient) through the transposed kernel
template <int D, typename T>
void launch_softmin_bwd_t32(const SoftminParams<T>& prm, const Ranges& rg, int n_ranges, int B, int N, int M, const Scratch& sc,
                            hipStream_t st) {
    WsumParams<T> w;
    w.x = prm.x; w.y = prm.y; w.s = prm.h; w.fwd = prm.fwd; w.g = prm.g; w.out = prm.out; w.gx = prm.gx;
    w.s2 = prm.s2; w.out_scale = prm.out_scale; w.gscale = 1.f; w.tscale = prm.shift2;
    launch_wsum_t32<WS_SOFTMIN_BWD, D, T, SoftminBwdOp<D, 2, false, 1, T>>(w, prm, rg, n_ranges, B, N, M, sc, st);
}

// gaussian gradient (FWDGRAD = false) or product + unit gradient (true) through the transposed kernel
template <int D, bool FWDGRAD, typename T>
void launch_gauss_grad_t32(const ConvParams<T>& prm, float blur, const Ranges& rg, int n_ranges, int B, int N, int M, const Scratch& sc,
                           hipStream_t st) {
    WsumParams<T> w;
    w.x = prm.x; w.y = prm.y; w.s = prm.v; w.fwd = nullptr; w.g = prm.g; w.out = prm.out; w.gx = prm.gx;
    w.s2 = kLog2e / (blur * blur); w.out_scale = 1.f; w.gscale = -1.0f / (blur * blur); w.tscale = prm.t;
    if constexpr (FWDGRAD) launch_wsum_t32<WS_GAUSS_FWDGRAD, D, T, GaussFwdGradMerge<D, T>>(w, prm, rg, n_ranges, B, N, M, sc, st);
    else launch_wsum_t32<WS_GAUSS_BWD, D, T, ConvOp<GLHIP_GAUSSIAN, D, 1, T, 1>>(w, prm, rg, n_ranges, B, N, M, sc, st);
}

#define GLHIP_XD_DISPATCH(D, CALL)                                                                                        \
    switch (D) {                                                                                                          \
        case 4: CALL(4); break;   case 5: CALL(5); break;   case 6: CALL(6); break;   case 7: CALL(7); break;              \
        case 8: CALL(8); break;   case 9: CALL(9); break;   case 10: CALL(10); break; case 11: CALL(11); break;            \
        case 12: CALL(12); break; case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break;            \
        default: CALL(16); break;                                                                                         \
    }

inline bool use_mfma_dist(int flags, int n_ranges, int B, int D) {
    return (flags & GLHIP_FLAG_MFMA_DIST) != 0 && n_ranges > 0 && B == 1 && D <= 3;
}

template <int D, bool BWD, typename T>
void launch_softmin_d(const SoftminParams<T>& prm, const Ranges& rg, int n_ranges, int B, int N, int M, int p,
                      bool direct, bool mfma, int kind, const Scratch& sc, hipStream_t st) {
    if (p == 1) { launch_softmin_r<D, 1, true, BWD, T>(prm, rg, n_ranges, B, N, M, sc, st); return; }
    if (direct) { launch_softmin_r<D, 2, true, BWD, T>(prm, rg, n_ranges, B, N, M, sc, st); return; }
    if (!mfma) { launch_softmin_r<D, 2, false, BWD, T>(prm, rg, n_ranges, B, N, M, sc, st); return; }
    if constexpr (BWD) {
        launch_softmin_bwd_mfma<D, T>(prm, rg, n_ranges, B, N, M, sc, kind == FWD_X32, st);
    } else {   // `if constexpr`: the gradient translation unit does not instantiate the forward kernels, and vice versa
        // The two earlier tilings of the soft-min forward — fp32 MFMA (GLHIP_FLAG_F32_MFMA) and bf16 x 3 on 16x16x32 MFMAs
        // (GLHIP_FLAG_XDL16) — only ever served A/B runs against the shipped 32x32x16 kernel: they are compiled in with
        // `make AB=1` (-DGLHIP_AB_KERNELS) and left out of the shipped library, where the two flags select the default kernel.
#ifdef GLHIP_AB_KERNELS
        if (kind == FWD_XDL16) { launch_softmin_mfma<D, T, FWD_XDL16>(prm, rg, n_ranges, B, N, M, sc, st); return; }
        if (kind == FWD_F32) { launch_softmin_mfma<D, T, FWD_F32>(prm, rg, n_ranges, B, N, M, sc, st); return; }
#endif
        (void)kind;
        launch_softmin_mfma<D, T, FWD_X32>(prm, rg, n_ranges, B, N, M, sc, st);
    }
}

template <typename T>
SoftminParams<T> make_softmin_params(const void* x, const void* y, const float* h, float* out, float eps, int p,
                                     const float* pot, const float* prev, float alpha, float beta) {
    const float s2 = kLog2e / eps;
    SoftminParams<T> prm;
    prm.x = static_cast<const T*>(x);
    prm.y = static_cast<const T*>(y);
    prm.h = h;
    prm.out = out;
    prm.fwd = nullptr;
    prm.g = nullptr;
    prm.gx = nullptr;
    prm.s2 = s2;
    prm.t = (p == 1) ? s2 : std::sqrt(0.5f * s2);
    prm.inv_t = 1.0f / prm.t;
    prm.out_scale = -eps * kLn2;
    prm.clamp2 = 1e-8f * prm.t * prm.t;
    prm.pot = pot;
    prm.prev = prev;
    prm.pot_scale = 1.0f / eps;
    prm.alpha = alpha;
    prm.beta = beta;
    prm.shift2 = 0.f;
    return prm;
}

// glhip_sinkhorn_iter4: `count` dense p = 2 reductions in one launch of the x32 forward kernel + one merge launch
// rows x columns of one problem up to which an unbatched multi launch runs without column splits (5e6, measured)
static inline double tiny_multi_pairs() {
    constexpr double v = 5e6;
    return v;
}

template <typename T>
static inline int maxM_all(const SoftminMulti<T>& m) {
    int v = 0;
    for (int k = 0; k < m.count; ++k) v = m.M[k] > v ? m.M[k] : v;
    return v;
}

template <int D, typename T, int L = XL_BF16X3>
void launch_iter4(SoftminMulti<T>& m, int B, const Scratch& sc, hipStream_t st) {
    constexpr int NR = X32Layout<L>::NR;
    using MergeOp = SoftminFwdOp<D, 2, false, 1, T>;
    constexpr int NW = 4, kRows = NW * 32;
    int maxN = 0, minM = m.M[0];
    long row_blocks = 0;
    for (int k = 0; k < m.count; ++k) {
        maxN = m.N[k] > maxN ? m.N[k] : maxN;
        minM = m.M[k] < minM ? m.M[k] : minM;
        row_blocks += (long)B * ((m.N[k] + kRows - 1) / kRows);
    }
    const long per_split = (long)m.count * B * maxN * 2 * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits(row_blocks, minM, 0, fit) : 1;
    // ... with at least 3 column tiles per split: a 128-row workgroup that runs one tile of 512 columns is mostly prologue and
    // epilogue (N = M = 1e4, 19 splits by the rule: 64 us per iteration; 6: 57 us; 2e4: 173 -> 166 us; 5e3 and 3e4: unchanged)
    if (minM >= 3072 && minM / sp.n_splits < 1536) sp.n_splits = minM / 1536;
    // ... and none on tiny unbatched problems: the launch takes as long either way (N = M = 2000: 17.8 us with 3 splits + merge, 18.1 us
    // with one), and a loop of such launches is bound by the host's launch rate — the merge kernel is one launch in three
    if (B == 1 && (double)maxN * maxM_all(m) <= tiny_multi_pairs()) sp.n_splits = 1;
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = 0;   // per problem, set in the kernels
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    m.ws_stride = (long)sp.n_splits * B * maxN * 2;
    const int gx = (maxN + kRows - 1) / kRows;
    // Pre-packed columns (as in launch_softmin_mfma_nw): with 128-row workgroups every column is split into its bf16 pieces
    // (maxN / 128) times per problem; one more small launch does it once.  Measured (round 3): B x 4096^2 bf16 with B = 128 / 64 / 32
    // (the 2- / 4- / 8-GPU shards of configs[3]): 8.65 -> 8.28, 4.54 -> 4.35, 2.43 -> 2.32 ms per loss; N = M = 3e4: 4.07 -> 3.84 ms;
    // N = M = 1e4 and below: no difference (0.9 ms).
    double pairs = 0.0;
    int maxM = 0;
    for (int k = 0; k < m.count; ++k) {
        pairs += (double)B * m.N[k] * m.M[k];
        maxM = m.M[k] > maxM ? m.M[k] : maxM;
        m.pk[k] = PackedCols{nullptr, (long)((m.M[k] + 31) / 32) * (32 * NR)};
    }
    constexpr double pre_min = 1e8;
    bool pre = sc.ws && sc.allow_split && pairs >= pre_min;
    if (pre) {
        size_t off = (((size_t)(sp.n_splits > 1 ? m.count : 0) * (size_t)m.ws_stride * sizeof(float)) + 255) & ~(size_t)255;
        for (int k = 0; k < m.count && pre; ++k) {
            const size_t bytes = (size_t)B * (size_t)m.pk[k].stride * sizeof(uint4);
            if (off + bytes > sc.bytes) { pre = false; break; }
            m.pk[k].rec = reinterpret_cast<uint4*>(static_cast<char*>(sc.ws) + off);
            off += (bytes + 255) & ~(size_t)255;
        }
    }
    if (pre) {
        hipLaunchKernelGGL((pack_columns_multi_kernel<D, T, L>), dim3((maxM + 31 + kBlock) / kBlock, B, m.count), dim3(kBlock), 0, st, m);
        hipLaunchKernelGGL((softmin_fwd_x32_multi_kernel<D, T, NW, true, L>), dim3(gx, B, sp.n_splits * m.count), dim3(NW * 64), 0, st, m, sp);
    } else {
        hipLaunchKernelGGL((softmin_fwd_x32_multi_kernel<D, T, NW, false, L>), dim3(gx, B, sp.n_splits * m.count), dim3(NW * 64), 0, st, m, sp);
    }
    if (sp.n_splits > 1)
        hipLaunchKernelGGL((merge_multi_kernel<MergeOp, T>), dim3((maxN + kBlock - 1) / kBlock, B, m.count), dim3(kBlock), 0, st, m, sp);
}

// the same for 4 <= D <= 16: the multi launch of the xd kernel (glhip_softmin_xd.h), columns packed on the fly, 4 wavefronts x 1 row tile
template <int D, typename T, int L>
void launch_iter4_xd(SoftminMulti<T>& m, int B, const Scratch& sc, hipStream_t st) {
    using MergeOp = SoftminFwdOp<D, 2, false, 1, T>;
    constexpr int NW = 4, kRows = NW * 32;
    int maxN = 0, minM = m.M[0];
    long row_blocks = 0;
    for (int k = 0; k < m.count; ++k) {
        maxN = m.N[k] > maxN ? m.N[k] : maxN;
        minM = m.M[k] < minM ? m.M[k] : minM;
        row_blocks += (long)B * ((m.N[k] + kRows - 1) / kRows);
        m.pk[k] = PackedCols{nullptr, 0};
    }
    const long per_split = (long)m.count * B * maxN * 2 * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits(row_blocks, minM, 0, fit) : 1;
    // (no tiny-launch rule as in launch_iter4: N = M = 2000, D = 4 / 16 run 0.35 / 0.42 ms per loss with 3 splits, 0.43 / 0.58 ms with one)
    // (and more splits than the rule's do not pay either: 7 splits of 256 columns at N = M = 2000 measure like 3 — these loops are host-bound)
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = 0;   // per problem, set in the kernels
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    m.ws_stride = (long)sp.n_splits * B * maxN * 2;
    const int gx = (maxN + kRows - 1) / kRows;
    hipLaunchKernelGGL((xd_fwd_multi_kernel<D, T, NW, L>), dim3(gx, B, sp.n_splits * m.count), dim3(NW * 64), 0, st, m, sp);
    if (sp.n_splits > 1)
        hipLaunchKernelGGL((merge_multi_kernel<MergeOp, T>), dim3((maxN + kBlock - 1) / kBlock, B, m.count), dim3(kBlock), 0, st, m, sp);
}

// p = 1 (round 5): the multi launch of the dense distance kernel (glhip_dist_xd.h), any D <= 16
template <int D, typename T>
void launch_iter4_dist(SoftminMulti<T>& m, int B, float eps, const Scratch& sc, hipStream_t st) {
    using MergeOp = SoftminFwdOp<D, 1, true, 1, T>;
    constexpr int NW = 8, kRows = NW * 32;
    const float s2 = kLog2e / eps;
    DistMulti<T> dm;
    dm.count = m.count;
    int maxN = 0, minM = m.M[0];
    long row_blocks = 0;
    for (int k = 0; k < 4; ++k) {
        const SoftminParams<T>& q = m.p[k];
        dm.p[k] = DistParams<T>{q.x, q.y, q.h, q.pot, q.prev, q.out, s2, 1e-8f * s2 * s2, q.out_scale, q.pot_scale, q.alpha, q.beta, dist_guard()};
        dm.N[k] = m.N[k];
        dm.M[k] = m.M[k];
        if (k >= m.count) continue;
        maxN = m.N[k] > maxN ? m.N[k] : maxN;
        minM = m.M[k] < minM ? m.M[k] : minM;
        row_blocks += (long)B * ((m.N[k] + kRows - 1) / kRows);
    }
    const long per_split = (long)m.count * B * maxN * 2 * sizeof(float);
    const long fit = sc.ws ? (long)(sc.bytes / per_split) : 0;
    SplitInfo sp;
    sp.n_splits = (sc.allow_split && fit >= 2) ? choose_splits(row_blocks, minM, 0, fit) : 1;
    // (no tiny-launch rule here: p = 1 at N = M = 2000 runs 0.54 ms per loss with 3 splits, 0.84 ms with one) — the other way round:
    sp.n_splits = dist_small_launch_splits(sp.n_splits, row_blocks, minM, fit, sc.allow_split);
    sp.workspace = static_cast<float*>(sc.ws);
    sp.split_stride = 0;
    sp.xcd_grid_x = 0;
    sp.xcd_blocks = 0;
    m.ws_stride = dm.ws_stride = (long)sp.n_splits * B * maxN * 2;
    const int gx = (maxN + kRows - 1) / kRows;
    hipLaunchKernelGGL((dist_xd_multi_kernel<D, T, NW>), dim3(gx, B, sp.n_splits * m.count), dim3(NW * 64), 0, st, dm, sp);
    if (sp.n_splits > 1)
        hipLaunchKernelGGL((merge_multi_kernel<MergeOp, T>), dim3((maxN + kBlock - 1) / kBlock, B, m.count), dim3(kBlock), 0, st, m, sp);
}

// every problem of a SoftminMulti in one launch, on the kernel family of (p, D, layout)
template <typename T>
int multi_dispatch(SoftminMulti<T>& m, int B, int D, float eps, int p, const Scratch& sc, hipStream_t st) {
    if (m.count < 4) {      // unused slots: no rows
        for (int k = m.count; k < 4; ++k) { m.p[k] = m.p[0]; m.N[k] = 0; m.M[k] = m.M[0]; }
    }
    if (p == 1) {      // distances: one kernel for every D <= 16
        if (D == 1) launch_iter4_dist<1, T>(m, B, eps, sc, st);
        else if (D == 2) launch_iter4_dist<2, T>(m, B, eps, sc, st);
        else if (D == 3) launch_iter4_dist<3, T>(m, B, eps, sc, st);
        else {
#define GL_XD(DD) launch_iter4_dist<DD, T>(m, B, eps, sc, st)
            GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
        }
        return GLHIP_OK;
    }
    if (D > 3) {      // 4 <= D <= 16 (round 5)
#define GL_XD(DD) \
    if (sc.h2) launch_iter4_xd<DD, T, XL_F16X2>(m, B, sc, st); \
    else launch_iter4_xd<DD, T, XL_BF16X3>(m, B, sc, st)
        GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
        return GLHIP_OK;
    }
    if (sc.h2) {      // GLHIP_FLAG_F16X2: the iteration on the f16 x 2 layout, like the half-steps it replaces
        if (D == 1) launch_iter4<1, T, XL_F16X2>(m, B, sc, st);
        else if (D == 2) launch_iter4<2, T, XL_F16X2>(m, B, sc, st);
        else launch_iter4<3, T, XL_F16X2>(m, B, sc, st);
        return GLHIP_OK;
    }
    if (D == 1) launch_iter4<1, T>(m, B, sc, st);
    else if (D == 2) launch_iter4<2, T>(m, B, sc, st);
    else launch_iter4<3, T>(m, B, sc, st);
    return GLHIP_OK;
}

template <typename T>
int iter4_typed(const void* x, const void* y, const float* a_log, const float* b_log, const float* f_ba, const float* g_ab,
                const float* f_aa, const float* g_bb, float* f_ba_out, float* g_ab_out, float* f_aa_out, float* g_bb_out,
                int B, int N, int M, int D, float eps, float damping, int p, int first, const Scratch& sc, hipStream_t st) {
    // first = 0: averaged update;  1: initial potentials (no pot, no prev);  2: plain extrapolation (pot, no prev)
    const float alpha = first ? damping : 0.5f * damping, beta = 0.5f;
    auto one = [&](const void* rows, const void* cols, const float* logw, const float* pot, const float* prev, float* out) {
        return make_softmin_params<T>(rows, cols, logw, out, eps, p, first == 1 ? nullptr : pot, first ? nullptr : prev, alpha, beta);
    };
    SoftminMulti<T> m;
    m.count = f_aa_out ? 4 : 2;
    m.p[0] = one(x, y, b_log, g_ab, f_ba, f_ba_out); m.N[0] = N; m.M[0] = M;
    m.p[1] = one(y, x, a_log, f_ba, g_ab, g_ab_out); m.N[1] = M; m.M[1] = N;
    if (m.count == 4) {
        m.p[2] = one(x, x, a_log, f_aa, f_aa, f_aa_out); m.N[2] = N; m.M[2] = N;
        m.p[3] = one(y, y, b_log, g_bb, g_bb, g_bb_out); m.N[3] = M; m.M[3] = M;
    }
    return multi_dispatch<T>(m, B, D, eps, p, sc, st);
}

// The coarse-to-fine jump of the two-scale loop: every potential carried from the coarse measures to the fine points,
//     f_ba(x_i) = damping * softmin(eps, C(x_i, y_c), b_log_c + g_ab_c / eps)   and its three companions,
// as ONE launch of the same multi kernels (rows: fine clouds, columns: coarse clouds).
template <typename T>
int extrapolate4_typed(const void* x, const void* y, const void* xc, const void* yc, const float* a_log_c, const float* b_log_c,
                       const float* f_ba, const float* g_ab, const float* f_aa, const float* g_bb, float* f_ba_out, float* g_ab_out,
                       float* f_aa_out, float* g_bb_out, int B, int N, int M, int Nc, int Mc, int D, float eps, float damping, int p,
                       const Scratch& sc, hipStream_t st) {
    auto one = [&](const void* rows, const void* cols, const float* logw, const float* pot, float* out) {
        return make_softmin_params<T>(rows, cols, logw, out, eps, p, pot, nullptr, damping, 0.5f);
    };
    SoftminMulti<T> m;
    m.count = f_aa_out ? 4 : 2;
    m.p[0] = one(x, yc, b_log_c, g_ab, f_ba_out); m.N[0] = N; m.M[0] = Mc;
    m.p[1] = one(y, xc, a_log_c, f_ba, g_ab_out); m.N[1] = M; m.M[1] = Nc;
    if (m.count == 4) {
        m.p[2] = one(x, xc, a_log_c, f_aa, f_aa_out); m.N[2] = N; m.M[2] = Nc;
        m.p[3] = one(y, yc, b_log_c, g_bb, g_bb_out); m.N[3] = M; m.M[3] = Mc;
    }
    return multi_dispatch<T>(m, B, D, eps, p, sc, st);
}

struct StepArgs {   // fused Sinkhorn half-step; all-default = plain soft-min
    const float* pot = nullptr;
    const float* prev = nullptr;
    float alpha = 1.f, beta = 0.f;
    float shift2 = 0.f;     // value-and-gradient mode of the gradient kernels
};

template <bool BWD, typename T>
int softmin_typed(const void* x, const void* y, const float* h, float* out, const float* fwd, const float* g,
                  float* gx, int B, int N, int M, int D, float eps, int p, const Ranges& rg, int n_ranges,
                  const Scratch& sc, int flags, hipStream_t st, const StepArgs& step = StepArgs()) {
    const float s2 = kLog2e / eps;
    const float out_scale = -eps * kLn2;
    if (D <= 3) {
        const bool direct = (flags & GLHIP_FLAG_DIRECT) != 0;
        SoftminParams<T> prm;
        prm.x = static_cast<const T*>(x);
        prm.y = static_cast<const T*>(y);
        prm.h = h;
        prm.out = out;
        prm.fwd = fwd;
        prm.g = g;
        prm.gx = gx;
        prm.s2 = s2;
        prm.t = (p == 1) ? s2 : std::sqrt(0.5f * s2);
        prm.inv_t = 1.0f / prm.t;
        prm.out_scale = out_scale;
        prm.clamp2 = 1e-8f * prm.t * prm.t;
        prm.pot = step.pot;
        prm.prev = step.prev;
        prm.pot_scale = 1.0f / eps;
        prm.alpha = step.alpha;
        prm.beta = step.beta;
        prm.shift2 = step.shift2;
        const bool mfma = (flags & GLHIP_FLAG_NO_MFMA) == 0;
        const int xdl = (flags & GLHIP_FLAG_F32_MFMA) ? FWD_F32 : (flags & GLHIP_FLAG_XDL16) ? FWD_XDL16 : FWD_X32;
        if constexpr (!BWD) {
            if (p == 1 && use_mfma_dist(flags, n_ranges, B, D)) {
                DistParams<T> dp{prm.x, prm.y, h, step.pot, step.prev, out, s2, 1e-8f * s2 * s2, out_scale, 1.0f / eps, step.alpha, step.beta, dist_guard()};
                if (D == 1) launch_dist<DM_SOFTMIN_P1, 1, T, SoftminFwdOp<1, 1, true, 1, T>>(dp, prm, rg, n_ranges, N, M, sc, st);
                else if (D == 2) launch_dist<DM_SOFTMIN_P1, 2, T, SoftminFwdOp<2, 1, true, 1, T>>(dp, prm, rg, n_ranges, N, M, sc, st);
                else launch_dist<DM_SOFTMIN_P1, 3, T, SoftminFwdOp<3, 1, true, 1, T>>(dp, prm, rg, n_ranges, N, M, sc, st);
                return GLHIP_OK;
            }
        }
        if constexpr (!BWD) {
            // GLHIP_FLAG_F16X2, D <= 3: 3 D + 6 <= 15 K slots = ONE v_mfma_f32_32x32x16_f16 per 1024 exponents (bf16 x 3: two), 32 bytes
            // of LDS per column (64) — the kernel of glhip_softmin_xd.h instantiated for D <= 3
            // Big dense launches only (pre-packed columns, XCD-aware grid): that is where it was measured to win — 87.1 -> 79.2 ms at
            // 1e6 x 1e6, 36.8 -> 35.0 ms per online loss at 1e5; batches of 4096 x 4096 problems lose 4 % to the x32 kernel's staging
            // (B = 256: 16.2 -> 17.0 ms per loss), and block-sparse launches keep the gathered pre-packed tiles of glhip_softmin_x32.h.
            constexpr bool via_xd = true;
            // (block-sparse launches through this kernel, packing their tiles on the fly in 512-row workgroups: 281 vs 244 ms per two-scale
            // loss at 1e6, round 6)
            // (mid-size launches, 16384 <= M < 65536, through this kernel with pre-packed columns and a free split count: 0.164 -> 0.191 ms
            // at N = M = 4e4, 0.242 -> 0.250 at 5e4, round 6: they stay on the x32 kernel)
            if (via_xd && p == 2 && sc.h2 && !direct && mfma && xdl == FWD_X32 && n_ranges == 0 && M >= 65536 && (double)B * N * M >= 5e8) {
                if (D == 1) launch_xd_l<XD_SOFTMIN, 1, T, SoftminFwdOp<1, 2, false, 1, T>, XL_F16X2>(prm, prm, rg, n_ranges, B, N, M, sc, st);
                else if (D == 2) launch_xd_l<XD_SOFTMIN, 2, T, SoftminFwdOp<2, 2, false, 1, T>, XL_F16X2>(prm, prm, rg, n_ranges, B, N, M, sc, st);
                else launch_xd_l<XD_SOFTMIN, 3, T, SoftminFwdOp<3, 2, false, 1, T>, XL_F16X2>(prm, prm, rg, n_ranges, B, N, M, sc, st);
                return GLHIP_OK;
            }
        }
        if (D == 1) launch_softmin_d<1, BWD, T>(prm, rg, n_ranges, B, N, M, p, direct, mfma, xdl, sc, st);
        else if (D == 2) launch_softmin_d<2, BWD, T>(prm, rg, n_ranges, B, N, M, p, direct, mfma, xdl, sc, st);
        else launch_softmin_d<3, BWD, T>(prm, rg, n_ranges, B, N, M, p, direct, mfma, xdl, sc, st);
    } else {
        if constexpr (!BWD) {
            if (p == 1 && D <= kXdMaxD && n_ranges == 0 && !(flags & (GLHIP_FLAG_NO_MFMA | GLHIP_FLAG_DIRECT))) {   // distances on the matrix cores
                const SoftminParams<T> mprm = make_softmin_params<T>(x, y, h, out, eps, 1, step.pot, step.prev, step.alpha, step.beta);
                const DistParams<T> dp{mprm.x, mprm.y, h, step.pot, step.prev, out, s2, 1e-8f * s2 * s2, out_scale, 1.0f / eps, step.alpha, step.beta,
                                       dist_guard()};
#define GL_XD(DD) launch_dist_xd<DM_SOFTMIN_P1, DD, T, SoftminFwdOp<DD, 1, true, 1, T>>(dp, mprm, B, N, M, sc, st)
                GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
                return GLHIP_OK;
            }
            if (p == 2 && D <= kXdMaxD && !(flags & (GLHIP_FLAG_NO_MFMA | GLHIP_FLAG_DIRECT))) {   // 4 <= D <= 16: matrix cores
                SoftminParams<T> prm = make_softmin_params<T>(x, y, h, out, eps, 2, step.pot, step.prev, step.alpha, step.beta);
#define GL_XD(DD) launch_xd<XD_SOFTMIN, DD, T, SoftminFwdOp<DD, 2, false, 1, T>>(prm, prm, rg, n_ranges, B, N, M, sc, st)
                GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
                return GLHIP_OK;
            }
        } else {
            if (p == 1 && D <= kXdMaxD && n_ranges == 0 && !out && !(flags & (GLHIP_FLAG_NO_MFMA | GLHIP_FLAG_DIRECT))) {   // p = 1 gradient, dense
                SoftminParams<T> mprm = make_softmin_params<T>(x, y, h, nullptr, eps, 1, nullptr, nullptr, 1.f, 0.f);
                mprm.fwd = fwd; mprm.g = g; mprm.gx = gx;
                DistXdGradParams<T> gp;
                gp.d = DistParams<T>{mprm.x, mprm.y, h, nullptr, nullptr, nullptr, s2, 1e-8f * s2 * s2, out_scale, 0.f, 1.f, 0.f, dist_guard()};
                gp.fwd = fwd; gp.g = g; gp.gx = gx; gp.gscale = 1.f;
#define GL_XD(DD) launch_dist_xd_grad<DM_SOFTMIN_P1, DD, T, SoftminBwdOp<DD, 1, true, 1, T>>(gp, mprm, B, N, M, sc, st)
                GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
                return GLHIP_OK;
            }
            if (p == 2 && D <= kXdMaxD && !(flags & (GLHIP_FLAG_NO_MFMA | GLHIP_FLAG_DIRECT))) {   // gradient (+ value), 4 <= D <= 16
                SoftminParams<T> prm = make_softmin_params<T>(x, y, h, out, eps, 2, nullptr, nullptr, 1.f, 0.f);
                prm.fwd = fwd; prm.g = g; prm.gx = gx; prm.shift2 = step.shift2;
#define GL_XD(DD) launch_softmin_bwd_t32<DD, T>(prm, rg, n_ranges, B, N, M, sc, st)
                GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
                return GLHIP_OK;
            }
        }
        if (step.pot || step.prev || step.alpha != 1.f)
            return fail(GLHIP_EUNSUPPORTED, "glhip_sinkhorn_step: no fused kernel for D=%d, p=%d, flags=%d (D <= 16, p = 2 on the matrix "
                                            "cores only): use glhip_softmin_fwd", D, p, flags);
        if (BWD && D > kGenericMaxGradD)
            return fail(GLHIP_EUNSUPPORTED, "softmin_bwd_x: D=%d > %d is not supported by the generic gradient kernel",
                        D, kGenericMaxGradD);
        GenericParams<T> prm;
        prm.x = static_cast<const T*>(x);
        prm.y = static_cast<const T*>(y);
        prm.s = h;
        prm.out = out;
        prm.fwd = fwd;
        prm.g = g;
        prm.gx = gx;
        prm.dscale = (p == 1) ? s2 : 0.5f * s2;
        prm.out_scale = out_scale;
        prm.gscale = 1.f;
        prm.clamp2 = 1e-8f;
        const bool sp = n_ranges > 0;
        dim3 grid(sp ? n_ranges : (N + kBlock - 1) / kBlock, sp ? 1 : B, 1);
#define GL_LAUNCH(MODE, SP) \
    hipLaunchKernelGGL((generic_kernel<MODE, BWD, SP, T>), grid, dim3(kBlock), 0, st, prm, rg, N, M, D)
        if (p == 2) { if (sp) GL_LAUNCH(GM_SOFTMIN_P2, true); else GL_LAUNCH(GM_SOFTMIN_P2, false); }
        else        { if (sp) GL_LAUNCH(GM_SOFTMIN_P1, true); else GL_LAUNCH(GM_SOFTMIN_P1, false); }
#undef GL_LAUNCH
    }
    return GLHIP_OK;
}

// ---- kernel products ---------------------------------------------------------------------------------

template <int KIND, int D, int BWD, typename T>   // BWD: ConvOp's MODE (0 product, 1 gradient, 2 both)
void launch_conv_r(const ConvParams<T>& prm, const Ranges& rg, int n_ranges, int B, int N, int M, const Scratch& sc,
                   hipStream_t st) {
    if (use_two_rows(B, N, n_ranges, sc))
        launch_mapreduce<ConvOp<KIND, D, 2, T, BWD>>(prm, rg, n_ranges, B, N, M, sc.ws, sc.bytes, sc.allow_split, st, sc.cb);
    else
        launch_mapreduce<ConvOp<KIND, D, 1, T, BWD>>(prm, rg, n_ranges, B, N, M, sc.ws, sc.bytes, sc.allow_split, st, sc.cb);
}

template <int KIND, int BWD, typename T>
void launch_conv_d(const ConvParams<T>& prm, const Ranges& rg, int n_ranges, int B, int N, int M, int D,
                   const Scratch& sc, hipStream_t st) {
    if (D == 1) launch_conv_r<KIND, 1, BWD, T>(prm, rg, n_ranges, B, N, M, sc, st);
    else if (D == 2) launch_conv_r<KIND, 2, BWD, T>(prm, rg, n_ranges, B, N, M, sc, st);
    else launch_conv_r<KIND, 3, BWD, T>(prm, rg, n_ranges, B, N, M, sc, st);
}

template <int D, bool BWD, typename T>
void launch_gauss_mfma(const ConvParams<T>& prm, float blur, const Ranges& rg, int n_ranges, int B, int N, int M,
                       const Scratch& sc, bool x32, hipStream_t st) {
    WsumParams<T> w;
    w.x = prm.x; w.y = prm.y; w.s = prm.v; w.fwd = nullptr; w.g = prm.g; w.out = prm.out; w.gx = prm.gx;
    w.s2 = kLog2e / (blur * blur); w.out_scale = 1.f; w.gscale = -1.0f / (blur * blur); w.tscale = prm.t;
    if constexpr (BWD) launch_wsum<WS_GAUSS_BWD, D, T, ConvOp<GLHIP_GAUSSIAN, D, 1, T, true>>(w, prm, rg, n_ranges, B, N, M, sc, x32, st);
    else launch_wsum<WS_GAUSS_FWD, D, T, ConvOp<GLHIP_GAUSSIAN, D, 1, T, false>>(w, prm, rg, n_ranges, B, N, M, sc, x32, st);
}

// gaussian product + its row gradient in one pass (WS_GAUSS_FWDGRAD)
template <int D, typename T>
void launch_gauss_fwdgrad(const ConvParams<T>& prm, float blur, const Ranges& rg, int n_ranges, int B, int N, int M,
                          const Scratch& sc, hipStream_t st) {
    WsumParams<T> w;
    w.x = prm.x; w.y = prm.y; w.s = prm.v; w.fwd = nullptr; w.g = nullptr; w.out = prm.out; w.gx = prm.gx;
    w.s2 = kLog2e / (blur * blur); w.out_scale = 1.f; w.gscale = -1.0f / (blur * blur); w.tscale = prm.t;
    launch_wsum<WS_GAUSS_FWDGRAD, D, T, GaussFwdGradMerge<D, T>>(w, prm, rg, n_ranges, B, N, M, sc, false, st);
}

template <bool BWD, typename T>
int conv_typed(int kind, const void* x, const void* y, const float* v, float* out, const float* g, float* gx,
               int B, int N, int M, int D, float blur, const Ranges& rg, int n_ranges, const Scratch& sc,
               int flags, hipStream_t st) {
    if (D <= 3) {
        ConvParams<T> prm;
        prm.x = static_cast<const T*>(x);
        prm.y = static_cast<const T*>(y);
        prm.v = v;
        prm.out = out;
        prm.g = g;
        prm.gx = gx;
        if (kind == GLHIP_GAUSSIAN) {
            prm.t = std::sqrt(0.5f * kLog2e) / blur;
            prm.gscale = -1.0f / (prm.t * blur * blur);
            prm.clamp2 = 0.f;
            if ((flags & GLHIP_FLAG_NO_MFMA) == 0) {
                const bool x32 = (flags & GLHIP_FLAG_XDL16) == 0;
                if (D == 1) launch_gauss_mfma<1, BWD, T>(prm, blur, rg, n_ranges, B, N, M, sc, x32, st);
                else if (D == 2) launch_gauss_mfma<2, BWD, T>(prm, blur, rg, n_ranges, B, N, M, sc, x32, st);
                else launch_gauss_mfma<3, BWD, T>(prm, blur, rg, n_ranges, B, N, M, sc, x32, st);
            } else {
                launch_conv_d<GLHIP_GAUSSIAN, BWD, T>(prm, rg, n_ranges, B, N, M, D, sc, st);
            }
        } else if (kind == GLHIP_LAPLACIAN) {
            prm.t = kLog2e / blur;
            prm.gscale = -1.0f / blur;
            prm.clamp2 = 1e-8f * kLog2e * kLog2e;   // the reference clamps |x/blur - y/blur|^2
            if constexpr (!BWD) {
                if (use_mfma_dist(flags, n_ranges, B, D)) {      // GRAD_FAMILY: |.| = m rsq(m), as the product of glhip_dist_grad_x32.h
                    DistParams<T> dp{prm.x, prm.y, v, nullptr, nullptr, out, prm.t, prm.clamp2, 1.f, 0.f, 1.f, 0.f, dist_guard()};
                    const bool fam = (flags & GLHIP_FLAG_GRAD_FAMILY) != 0;
#define GL_DIST(DD) \
    if (fam) launch_dist<DM_LAPLACIAN, DD, T, ConvOp<GLHIP_LAPLACIAN, DD, 1, T, false>, true>(dp, prm, rg, n_ranges, N, M, sc, st); \
    else launch_dist<DM_LAPLACIAN, DD, T, ConvOp<GLHIP_LAPLACIAN, DD, 1, T, false>, false>(dp, prm, rg, n_ranges, N, M, sc, st)
                    if (D == 1) { GL_DIST(1); } else if (D == 2) { GL_DIST(2); } else { GL_DIST(3); }
#undef GL_DIST
                    return GLHIP_OK;
                }
                if (flags & GLHIP_FLAG_GRAD_FAMILY) {   // rounded like the product-and-gradient kernel (glhip_kconv_ops.h, MODE 3)
                    launch_conv_d<GLHIP_LAPLACIAN, 3, T>(prm, rg, n_ranges, B, N, M, D, sc, st);
                    return GLHIP_OK;
                }
            } else {
                if (use_mfma_dist(flags, n_ranges, B, D)) {
                    launch_dist_grad_d<GLHIP_LAPLACIAN, DG_BWD, T>(prm, rg, n_ranges, N, M, D, sc, st);
                    return GLHIP_OK;
                }
            }
            launch_conv_d<GLHIP_LAPLACIAN, BWD, T>(prm, rg, n_ranges, B, N, M, D, sc, st);
        } else {
            prm.t = 1.0f;
            prm.gscale = -1.0f;
            prm.clamp2 = 1e-8f;
            if constexpr (!BWD) {
                if (use_mfma_dist(flags, n_ranges, B, D)) {
                    DistParams<T> dp{prm.x, prm.y, v, nullptr, nullptr, out, 1.f, 1e-8f, 1.f, 0.f, 1.f, 0.f, dist_guard()};
                    const bool fam = (flags & GLHIP_FLAG_GRAD_FAMILY) != 0;
#define GL_DIST(DD) \
    if (fam) launch_dist<DM_ENERGY, DD, T, ConvOp<GLHIP_ENERGY, DD, 1, T, false>, true>(dp, prm, rg, n_ranges, N, M, sc, st); \
    else launch_dist<DM_ENERGY, DD, T, ConvOp<GLHIP_ENERGY, DD, 1, T, false>, false>(dp, prm, rg, n_ranges, N, M, sc, st)
                    if (D == 1) { GL_DIST(1); } else if (D == 2) { GL_DIST(2); } else { GL_DIST(3); }
#undef GL_DIST
                    return GLHIP_OK;
                }
                if (flags & GLHIP_FLAG_GRAD_FAMILY) {   // rounded like the product-and-gradient kernel (glhip_kconv_ops.h, MODE 3)
                    launch_conv_d<GLHIP_ENERGY, 3, T>(prm, rg, n_ranges, B, N, M, D, sc, st);
                    return GLHIP_OK;
                }
            } else {
                if (use_mfma_dist(flags, n_ranges, B, D)) {
                    launch_dist_grad_d<GLHIP_ENERGY, DG_BWD, T>(prm, rg, n_ranges, N, M, D, sc, st);
                    return GLHIP_OK;
                }
            }
            launch_conv_d<GLHIP_ENERGY, BWD, T>(prm, rg, n_ranges, B, N, M, D, sc, st);
        }
    } else {
        if constexpr (!BWD) {
            if (kind != GLHIP_GAUSSIAN && D <= kXdMaxD && n_ranges == 0 && !(flags & GLHIP_FLAG_NO_MFMA)) {   // laplacian / energy: distances on the matrix cores
                const bool lap = kind == GLHIP_LAPLACIAN;
                const float t = lap ? kLog2e / blur : 1.0f;
                ConvParams<T> mprm;
                mprm.x = static_cast<const T*>(x); mprm.y = static_cast<const T*>(y); mprm.v = v; mprm.out = out; mprm.g = nullptr; mprm.gx = nullptr;
                mprm.t = t; mprm.gscale = 0.f; mprm.clamp2 = 1e-8f * (lap ? kLog2e * kLog2e : 1.f);
                const DistParams<T> dp{mprm.x, mprm.y, v, nullptr, nullptr, out, t, mprm.clamp2, 1.f, 0.f, 1.f, 0.f, dist_guard()};
#define GL_XD(DD) \
    if (lap) launch_dist_xd<DM_LAPLACIAN, DD, T, ConvOp<GLHIP_LAPLACIAN, DD, 1, T, 0>>(dp, mprm, B, N, M, sc, st); \
    else launch_dist_xd<DM_ENERGY, DD, T, ConvOp<GLHIP_ENERGY, DD, 1, T, 0>>(dp, mprm, B, N, M, sc, st)
                GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
                return GLHIP_OK;
            }
            if (kind == GLHIP_GAUSSIAN && D <= kXdMaxD && !(flags & GLHIP_FLAG_NO_MFMA)) {   // 4 <= D <= 16: matrix cores
                // the gaussian exponent -|x-y|^2 / (2 blur^2) is the soft-min's with eps = blur^2 and h = 0; `h` carries v
                const SoftminParams<T> prm = make_softmin_params<T>(x, y, v, out, blur * blur, 2, nullptr, nullptr, 1.f, 0.f);
                ConvParams<T> mprm;
                mprm.x = prm.x; mprm.y = prm.y; mprm.v = v; mprm.out = out; mprm.g = nullptr; mprm.gx = nullptr;
                mprm.t = 1.f; mprm.gscale = 0.f; mprm.clamp2 = 0.f;
#define GL_XD(DD) launch_xd<XD_GAUSS, DD, T, ConvOp<GLHIP_GAUSSIAN, DD, 1, T, 0>>(prm, mprm, rg, n_ranges, B, N, M, sc, st)
                GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
                return GLHIP_OK;
            }
        } else {
            if (kind != GLHIP_GAUSSIAN && D <= kXdMaxD && n_ranges == 0 && !(flags & GLHIP_FLAG_NO_MFMA)) {   // laplacian / energy gradients, dense
                const bool lap = kind == GLHIP_LAPLACIAN;
                const float t = lap ? kLog2e / blur : 1.0f;
                ConvParams<T> mprm;
                mprm.x = static_cast<const T*>(x); mprm.y = static_cast<const T*>(y); mprm.v = v; mprm.out = nullptr; mprm.g = g; mprm.gx = gx;
                mprm.t = t; mprm.gscale = lap ? -1.0f / blur : -1.0f; mprm.clamp2 = 1e-8f * (lap ? kLog2e * kLog2e : 1.f);
                DistXdGradParams<T> gp;
                gp.d = DistParams<T>{mprm.x, mprm.y, v, nullptr, nullptr, nullptr, t, mprm.clamp2, 1.f, 0.f, 1.f, 0.f, dist_guard()};
                gp.fwd = nullptr; gp.g = g; gp.gx = gx; gp.gscale = mprm.gscale;
#define GL_XD(DD) \
    if (lap) launch_dist_xd_grad<DM_LAPLACIAN, DD, T, ConvOp<GLHIP_LAPLACIAN, DD, 1, T, 1>>(gp, mprm, B, N, M, sc, st); \
    else launch_dist_xd_grad<DM_ENERGY, DD, T, ConvOp<GLHIP_ENERGY, DD, 1, T, 1>>(gp, mprm, B, N, M, sc, st)
                GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
                return GLHIP_OK;
            }
            if (kind == GLHIP_GAUSSIAN && D <= kXdMaxD && !(flags & GLHIP_FLAG_NO_MFMA)) {   // gaussian gradient, 4 <= D <= 16
                ConvParams<T> prm;
                prm.x = static_cast<const T*>(x); prm.y = static_cast<const T*>(y); prm.v = v; prm.out = out; prm.g = g; prm.gx = gx;
                prm.t = std::sqrt(0.5f * kLog2e) / blur;
                prm.gscale = -1.0f / (prm.t * blur * blur);
                prm.clamp2 = 0.f;
#define GL_XD(DD) launch_gauss_grad_t32<DD, false, T>(prm, blur, rg, n_ranges, B, N, M, sc, st)
                GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
                return GLHIP_OK;
            }
        }
        if (BWD && D > kGenericMaxGradD)
            return fail(GLHIP_EUNSUPPORTED, "kernel_conv_bwd_x: D=%d > %d is not supported by the generic gradient kernel",
                        D, kGenericMaxGradD);
        GenericParams<T> prm;
        prm.x = static_cast<const T*>(x);
        prm.y = static_cast<const T*>(y);
        prm.s = v;
        prm.out = out;
        prm.fwd = nullptr;
        prm.g = g;
        prm.gx = gx;
        prm.out_scale = 1.f;
        prm.clamp2 = 1e-8f;
        const bool sp = n_ranges > 0;
        dim3 grid(sp ? n_ranges : (N + kBlock - 1) / kBlock, sp ? 1 : B, 1);
#define GL_LAUNCH(MODE, SP) \
    hipLaunchKernelGGL((generic_kernel<MODE, BWD, SP, T>), grid, dim3(kBlock), 0, st, prm, rg, N, M, D)
        if (kind == GLHIP_GAUSSIAN) {
            prm.dscale = 0.5f * kLog2e / (blur * blur);
            prm.gscale = -1.0f / (blur * blur);
            if (sp) GL_LAUNCH(GM_GAUSS, true); else GL_LAUNCH(GM_GAUSS, false);
        } else if (kind == GLHIP_LAPLACIAN) {
            prm.dscale = kLog2e / blur;
            prm.gscale = -1.0f / blur;
            prm.clamp2 = 1e-8f * blur * blur;
            if (sp) GL_LAUNCH(GM_LAPLACE, true); else GL_LAUNCH(GM_LAPLACE, false);
        } else {
            prm.dscale = 1.f;
            prm.gscale = -1.0f;
            if (sp) GL_LAUNCH(GM_ENERGY, true); else GL_LAUNCH(GM_ENERGY, false);
        }
#undef GL_LAUNCH
    }
    return GLHIP_OK;
}

}  // namespace
