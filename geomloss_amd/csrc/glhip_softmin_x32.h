// glhip_softmin_x32.h — soft-min forward (p = 2, D <= 3), bf16x3 exponents on v_mfma_f32_32x32x16_bf16.
//
// Same arithmetic as glhip_softmin_xdl.h (every fp32 operand = 3 exact bf16 pieces, 8 products per coordinate),
// different tiling, chosen from measurements on MI355X (tools/ubench/overlap.hip, profiles/r01_ubench_pipes.txt):
// beside exp2-bound VALU work a 16x16x32 MFMA still costs ~6 issue cycles per 256 exponents, a 32x32x16 MFMA
// costs ~0 per 1024 — the matrix pipe time itself hides completely.  So:
//   * one MFMA pair (K = 2 x 16) produces a 32 x 32 block of exponents;
//   * the block is TRANSPOSED with respect to the 16x16 kernel: the MFMA "A" rows are 32 columns y_j (from LDS),
//     the MFMA "B" columns are 32 rows x_i (in registers).  In the result lane l owns row i = l % 32 and its 16
//     registers are 16 different columns, so the row sum is 15 adds inside the lane + one add into ONE running
//     register (not a 4- or 16-register tile), the running max is one value per lane, and the final merge of a
//     row is a single cross-half shuffle;
//   * the running max enters through the K slots instead of the accumulator input: K layout (4 blocks of 8 slots,
//     lanes 0-31 hold blocks 0 / 2, lanes 32-63 blocks 1 / 3 of the two MFMAs)
//       block d < D : y side [y1,y2,y1,y3,y1,y2,y3,y2]   x side [a1,a1,a2,a1,a3,a2,a2,a3]
//       block 3     : y side [H1,H2,H3, 1, 1, 1, 0, 0]   x side [ 1, 1, 1,n1,n2,n3, 0, 0]     n = -running max
//     and C = 0 (an inline constant: no accumulator-input registers at all).
// Lazy max, speculative tile pass, tile-end check, exact redo, column splits and the merge are as in the 16x16 kernel.
#pragma once

#include "glhip_klayout.h"

namespace glhip {


__device__ __forceinline__ f32x16 mfma_x32(const uint4& a, const uint4& b, const f32x16& c) {
    Pack16 pa, pb;
    pa.u = a;
    pb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.v, pb.v, c, 0, 0, 0);
}

__device__ __forceinline__ uint4 pack_h1(float h) {    // [H1,H2,H3,1,1,1,0,0]
    uint32_t p1, p2, p3;
    split3(h, p1, p2, p3);
    return uint4{p1 | (p2 << 16), p3 | 0x3F800000u, 0x3F803F80u, 0u};
}
// x-side block 3, [1,1,1,n1,n2,n3,0,0] with n = -m
__device__ __forceinline__ uint4 pack_negmax(float m) {
    uint32_t p1, p2, p3;
    split3(-m, p1, p2, p3);
    return uint4{0x3F803F80u, 0x3F80u | (p1 << 16), p2 | (p3 << 16), 0u};
}

// lane-dependent choice between two 16-byte operands, field by field (a `?:` on the uint4 structs themselves is lowered through
// scratch memory: both values stored, one re-loaded at a lane-dependent address)
__device__ __forceinline__ uint4 select_u4(bool c, const uint4& a, const uint4& b) {
    return uint4{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w};
}

__device__ __forceinline__ float max16(const f32x16& v) {
    const float a = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), b = fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]));
    const float c = fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])), d = fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float sum_exp2_16(const f32x16& v) {
    float e[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) e[k] = fast_exp2(v[k]);
    return (((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]))) +
           (((e[8] + e[9]) + (e[10] + e[11])) + ((e[12] + e[13]) + (e[14] + e[15])));
}
__device__ __forceinline__ float sum_exp2_16(const f32x16& v, float m) {
    float e[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) e[k] = fast_exp2(v[k] - m);
    return (((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]))) +
           (((e[8] + e[9]) + (e[10] + e[11])) + ((e[12] + e[13]) + (e[14] + e[15])));
}

// One column of the cost matrix as the four 16-byte MFMA records of the layout above (coordinates relative to
// `centre`), written `stride` records apart starting at `base` (K block 0).
// records per column: 4 (three coordinate blocks + the scalar block) in the bf16 x 3 layout above; 2 in the f16 x 2 layout of
// glhip_klayout.h (GLHIP_FLAG_F16X2: 3 D + 6 <= 15 K slots — ONE MFMA per 32 x 32 block, half the LDS and packed-column bytes)
template <int L> struct X32Layout { static constexpr int NR = (L == XL_F16X2) ? 2 : 4; };

template <int D, typename T, int L = XL_BF16X3>
__device__ __forceinline__ void pack_column(const SoftminParams<T>& prm, long col, bool valid, const float (&centre)[D],
                                            uint4* base, int stride) {
    float rec[4] = {0.f, 0.f, 0.f, kNegBig};
    if (valid) {
        float yj[D];
        load_point<D, T>(prm.y, col, yj);
        float n2 = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            rec[d] = yj[d] - centre[d];
            n2 = __builtin_fmaf(rec[d], rec[d], n2);
        }
        rec[3] = __builtin_fmaf(-0.5f * prm.s2, n2, dual_entry(prm, col) * kLog2e);
    }
    if constexpr (L == XL_F16X2) {      // both sides carry sqrt(s); exponents below the floor are the floor (glhip_klayout.h)
        const float q = __builtin_sqrtf(prm.s2);
        float ys[D];
#pragma unroll
        for (int d = 0; d < D; ++d) ys[d] = rec[d] * q;
        const float H = __builtin_fmaxf(rec[3], kH2Floor);
        base[0] = xd_record_of<D, false, L>(0, H, ys);
        base[stride] = xd_record_of<D, false, L>(1, H, ys);
        return;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) base[d * stride] = (d < D) ? pack_y(rec[d]) : uint4{0u, 0u, 0u, 0u};
    base[3 * stride] = pack_h1(rec[3]);
}

// Packed column records for a whole launch (PRE mode of the kernel below): launches with enough work split every
// column once here instead of once per row pass of every workgroup.  One centre per batch item (launch_centre: the mean of 8
// rows spread over it): for unsorted clouds a workgroup's own first row is a more arbitrary point than that; for
// cluster-sorted clouds (block-sparse mode) the per-workgroup centre of the on-the-fly path is more accurate, the global one
// has the accuracy of the dense launches (absolute error of a potential ~ 2^-24 diam^2, independent of eps).
struct PackedCols {
    uint4* rec;     // dense launches (GROUPED): [B][ceil(M/32)][NR K blocks][32 columns] (NR = 4, or 2 in the f16 x 2 layout) — the LDS tile layout, so a tile
                    //   (which starts on a group boundary there) is staged by a linear, fully coalesced copy;
                    // block-sparse launches: [M][4 K blocks] — tiles start at arbitrary columns
    long stride;    // GROUPED: records per batch item = ceil(M/32) * 32 NR
};

template <int D, typename T, bool GROUPED, int L = XL_BF16X3>
__global__ void __launch_bounds__(kBlock)
pack_columns_kernel(SoftminParams<T> prm, int N, int M, PackedCols pk) {
    constexpr int NR = X32Layout<L>::NR;
    const int b = blockIdx.y;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= (GROUPED ? ((M + 31) & ~31) : M)) return;
    float centre[D];
    launch_centre<D, T>(prm.x, b, N, centre);
    if (GROUPED) pack_column<D, T, L>(prm, (long)b * M + j, j < M, centre, pk.rec + b * pk.stride + (j >> 5) * (32 * NR) + (j & 31), 32);
    else pack_column<D, T, L>(prm, (long)b * M + j, true, centre, pk.rec + ((long)b * M + j) * NR, 1);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // register-friendly 16-byte value (HIP's uint4 is a struct)

// registers <- this thread's share of the next tile.  GROUPED: records r = tid, tid + THREADS, ... of the contiguous
// run `src` (cnt records); otherwise the 4 records of columns t = tid, tid + THREADS, ... (n real columns at `src`).
template <int PER, int THREADS, bool GROUPED, int NR = 4>
__device__ __forceinline__ void fetch_records(u32x4 (&pre)[PER], const uint4* src, int n, int tid) {
    if (GROUPED) {
        const int cnt = ((n + 31) & ~31) * NR;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int r = tid + k * THREADS;
            pre[k] = *reinterpret_cast<const u32x4*>(src + (r < cnt ? r : 0));
        }
    } else {
#pragma unroll
        for (int k = 0; k < PER / NR; ++k) {
            const int t = tid + k * THREADS;
            const u32x4* col = reinterpret_cast<const u32x4*>(src + (long)(t < n ? t : 0) * NR);
#pragma unroll
            for (int kb = 0; kb < NR; ++kb) pre[k * NR + kb] = col[kb];
        }
    }
}

// the next tile of columns of a workgroup: interval q of the block-sparse ranges (or the split's single interval),
// columns [j0, min(j0 + kTileX, je))
struct TileCursor {
    int q, j0, je;
};

template <bool SPARSE, bool ALIGN32>
__device__ __forceinline__ void open_interval(const Ranges& rg, int M, int q_end, int split, int ns, TileCursor& c) {
    // skip empty intervals; c.q == q_end (or beyond) means "no more tiles"
    while (c.q < q_end) {
        int js, je;
        if (!SPARSE && ALIGN32) {   // split boundaries on whole 32-column groups
            const int len = (((M + ns - 1) / ns) + 31) & ~31;
            js = min(M, split * len);
            je = min(M, js + len);
        } else {
            column_interval<SPARSE>(rg, M, c.q, split, ns, js, je);
        }
        if (js < je) {
            c.j0 = js;
            c.je = je;
            return;
        }
        c.q += SPARSE ? ns : 1;
    }
}

// Block-sparse launches with pre-packed columns GATHER their tiles: a tile is the next (up to) kTileX columns of the concatenation
// of the workgroup's column intervals, whatever their lengths, instead of one tile (stage + barrier + MFMAs + barrier) per piece
// of an interval.  The reference's cluster_scale rule makes ~2000 clusters whatever N is, so at N = 1e4 a row block of 5 points
// walks ~600 intervals of a few columns each (round 3: 156 us per soft-min against 10 us for the dense kernel on the same
// points), and at N = 1e6 (455 columns per cluster, runs of 1-3 kept clusters) a third of the tiles were short tails.
// gather_tile: the columns of slots t = tid + k * THREADS of the tile that starts at cursor c (-1: padding), the number of real
// columns of the tile, and the cursor after it.  The walk over the intervals is wave-uniform (scalar loads and loop control).
// `pieces`: how many interval pieces a tile may hold — 1 = one tile per piece (long intervals: the walk below would only delay
// the prefetch; measured 1 % at N = 1e6), kTileX = gather freely (short intervals).  The host decides per launch (SplitInfo::gather).
template <int K, int THREADS, int TILE = kTileX>
__device__ __forceinline__ TileCursor gather_tile(const Ranges& rg, int M, int q_end, int split, int ns, TileCursor c, int tid,
                                                  int (&cols)[K], int& n, int pieces) {
#pragma unroll
    for (int k = 0; k < K; ++k) cols[k] = -1;
    int off = 0;
    while (c.q < q_end && off < TILE && pieces-- > 0) {
        const int len = min(c.je - c.j0, TILE - off);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int t = tid + k * THREADS - off;
            if (t >= 0 && t < len) cols[k] = c.j0 + t;
        }
        off += len;
        c.j0 += len;
        if (c.j0 >= c.je) {
            c.q += ns;
            open_interval<true, true>(rg, M, q_end, split, ns, c);
        }
    }
    n = off;
    return c;
}

// Columns per LDS tile: 512 (32 KB) for workgroups of 4 / 8 wavefronts; 256 for the 2-wavefront workgroups that serve row blocks of
// up to 64 points (block-sparse launches on small clusters: twice the workgroups per CU for the same LDS, no idle wavefronts)
constexpr int fwd_tile(int NW) { return NW == 2 ? 256 : kTileX; }

// The work of one workgroup: row block bx of batch item b, column split `split`.  tileX: kTile * 4 records of LDS,
// [column group of 32][K block][column], one 16-byte record per (column, K block).
template <int D, typename T, bool SPARSE, int RT, int NW, bool PRE, int L = XL_BF16X3>
__device__ __forceinline__ void softmin_fwd_x32_body(const SoftminParams<T>& prm, const Ranges& rg, int N, int M,
                                                     const SplitInfo& sp, const PackedCols& pk, int bx, int b, int split,
                                                     uint4* tileX) {
    constexpr int kRowsPerWave = RT * 32;
    constexpr int kRowsPerBlock = NW * kRowsPerWave;
    constexpr int kThreads = NW * 64;
    constexpr int kTile = fwd_tile(NW);
    constexpr int NR = X32Layout<L>::NR;            // records per column
    constexpr bool H2 = (L == XL_F16X2);
    constexpr int GS = 32 * NR;                     // records per column group of 32
    constexpr int kPer = (kTile * NR) / kThreads;   // PRE: records one thread moves per tile
    static_assert((kTile * NR) % kThreads == 0, "tile / workgroup shape");
    constexpr float kFloor = H2 ? kH2Floor : kMinusHuge;      // the running maximum of a row that has seen no mass yet
    // Block-sparse, 4 wavefronts: a workgroup may carry ONE leftover row tile of its row block besides its own 4 (SplitInfo::share,
    // build_row_chunks_kernel): every wavefront reduces its own row tile against all the column groups of a tile and the leftover
    // row tile against every 4th group — 1.25 row tiles of work per wavefront, none of them idle.  (Up to 3 leftover tiles per
    // workgroup were tried: 22-25 more VGPRs per slot, 150 in all, 3 waves per SIMD.)
    constexpr int XS = (SPARSE && RT == 1 && NW == 4 && L == XL_F16X2) ? 1 : 0;      // (bf16 x 3: 94 -> 122 VGPRs and spills; left alone)
    constexpr int NS = RT + XS;                     // row-tile slots of a wavefront: [0, RT) its own, [RT, NS) the shared leftovers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps the loop control on the scalar unit
    const int ns = sp.n_splits;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int rec0 = half * 32 + l31;     // this lane's record inside a column group: K block `half` (+2 for the 2nd MFMA)

    int row_begin, row_end, q_begin, q_end;
    block_extent<SPARSE>(rg, N, kRowsPerBlock, row_begin, row_end, q_begin, q_end, bx);
    // share mode: a row block [r0, r1) of nt > 4 row tiles is cut into W = nt / 4 chunks of exactly 4 row tiles; of its last
    // rl = nt - 4 W row tiles, the first min(rl, W) go one each to workgroups of the block's chunks and the others form a trailing partial chunk
    int xb = 0, xe = 0;
    if constexpr (XS > 0) {
        if (sp.share && rg.chunks && rg.chunks[0] >= 0 && bx < rg.chunks[0]) {
            const int kb = rg.chunks[1 + 3 * bx];
            const int r0 = rg.ranges_i[2 * kb], r1 = rg.ranges_i[2 * kb + 1];
            const int nt = (r1 - r0 + 31) >> 5;
            if (nt > 4) {
                // which chunks carry is scattered with the row block's index: workgroups go round the 8 XCDs by their linear id, and
                // with "the first chunk of every block" equal blocks of 4 chunks sent all the 1.25-tile workgroups to XCDs 0 and 4
                // (tools/probe_sparse_ideal.py, 530-row clusters: 26.3 ms against 22.8 without carrying)
                const int W = nt >> 2, rl = nt - 4 * W, c = (row_begin - r0) >> 7;
                const int t = (c + W - (int)((((unsigned)kb * 0x9E3779B1u) >> 16) % (unsigned)W)) % W;      // leftover tile of chunk c, if < rl
                if (c < W && t < rl) {
                    xb = r0 + 128 * W + 32 * t;
                    xe = min(r1, xb + 32);
                }
            }
        }
    }
    const int nx = XS > 0 ? (xe - xb + 31) >> 5 : 0;      // leftover row tiles of this workgroup (wave-uniform; > 0 only beside a full chunk)

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint4 kOnes = uint4{0x3F803F80u, 0x00003F80u, 0u, 0u};   // [1,1,1,0,...]: block 3 with n = 0

    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerBlock) {
        float centre[D];
        if (PRE) launch_centre<D, T>(prm.x, b, N, centre);
        else load_point<D, T>(prm.x, (long)b * N + row0, centre);

        // block-sparse: the wavefront -> rows assignment is rotated by the chunk index (the empty slots of successive partial chunks
        // fall on different SIMDs), and a chunk that fills at most half of the workgroup's row tiles shares each row tile between
        // `cs` = 2 or 4 wavefronts, which take every cs-th column group of a tile and merge their (max, sum) pairs through LDS at the
        // end.  A cluster of 530 rows is 4 chunks of 128 + 18 rows: until round 6 those 18 rows cost a fifth full chunk — one
        // wavefront reducing, three staging and waiting — i.e. +25 % for 3.5 % more rows, on the ~30 % of the voxel clusters of a
        // uniform cloud that exceed a multiple of the chunk height (tools/probe_sparse_ideal.py: 416-row clusters 0.61 of the VALU
        // model against 0.76 for 384).
        const int wslot = SPARSE ? ((wave + bx) & (NW - 1)) : wave;
        int cs = 1;
        if (SPARSE && RT == 1 && NW >= 4) {
            const int tiles = (min(row_end, row0 + kRowsPerBlock) - row0 + 31) >> 5;      // row tiles of this pass
            cs = tiles * 4 <= NW ? 4 : (tiles * 2 <= NW ? 2 : 1);
        }
        const int cpart = wslot & (cs - 1);                // this wavefront's share of the column groups: cpart, cpart + cs, ...
        const int wave_row0 = row0 + (wslot / cs) * kRowsPerWave;
        uint4 Xlo[NS], Xhi[NS];
        float m[NS], ssum[NS];
#pragma unroll
        for (int rt = 0; rt < NS; ++rt) {
            if (rt >= RT && rt - RT >= nx) {      // no such leftover tile: the slot is never touched
                Xlo[rt] = Xhi[rt] = uint4{0u, 0u, 0u, 0u};
                m[rt] = kFloor;
                ssum[rt] = 0.f;
                continue;
            }
            const int i = rt < RT ? min(wave_row0 + rt * 32 + l31, row_end - 1) : min(xb + (rt - RT) * 32 + l31, xe - 1);
            float xi[D];
            load_point<D, T>(prm.x, (long)b * N + i, xi);
            if constexpr (H2) {      // one operand: record `half` of [k,k,k,n1,n2,n3 | a_hi,a_hi,a_lo per coordinate], n = 0 for now
                float a[D];
                const float q = __builtin_sqrtf(prm.s2);
#pragma unroll
                for (int d = 0; d < D; ++d) a[d] = (xi[d] - centre[d]) * q;
                Xlo[rt] = select_u4(half != 0, xd_record_of<D, true, L>(1, 0.f, a), xd_record_of<D, true, L>(0, 0.f, a));
                Xhi[rt] = Xlo[rt];
            } else {
                float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int d = 0; d < D; ++d) a[d] = (xi[d] - centre[d]) * prm.s2;
                const uint4 z = uint4{0u, 0u, 0u, 0u};
                const uint4 p0 = pack_a(a[0]), p1 = (D > 1) ? pack_a(a[1]) : z, p2 = (D > 2) ? pack_a(a[2]) : z;
                Xlo[rt] = half ? p1 : p0;
                Xhi[rt] = half ? kOnes : p2;
            }
            m[rt] = kFloor;
            ssum[rt] = 0.f;
        }
        const bool wave_active = wave_row0 < row_end;
        bool first_group = true, first_shared = true;

        // PRE: the records of the next tile are fetched into registers while the current tile is consumed
        u32x4 pre[kPer];
        // a padding column: zero coordinates, H = "minus infinity" of the layout
        u32x4 neutral[NR];
        {
            const float zeros[D] = {};
#pragma unroll
            for (int kb = 0; kb < NR; ++kb) {
                uint4 r = uint4{0u, 0u, 0u, 0u};
                if constexpr (H2) r = xd_record_of<D, false, L>(kb, kH2Floor, zeros);
                else if (kb == 3) r = pack_h1(kNegBig);
                neutral[kb] = u32x4{r.x, r.y, r.z, r.w};
            }
        }
        auto tile_src = [&](int j0) {   // first record of the tile starting at column j0 in the packed buffer
            return SPARSE ? pk.rec + ((long)b * M + j0) * NR : pk.rec + b * pk.stride + (long)(j0 >> 5) * GS;
        };
        TileCursor cur;
        cur.q = q_begin + (SPARSE ? split : 0);
        cur.j0 = cur.je = 0;
        open_interval<SPARSE, PRE>(rg, M, q_end, split, ns, cur);
        const int pieces = sp.gather ? kTile : 1;
        constexpr bool GATHER = SPARSE && PRE;   // pre-packed: the gathered tile is fetched into registers one tile ahead
        constexpr bool GATHER_NOW = SPARSE && !PRE;   // packed on the fly: gathered when it is staged
        constexpr int kCols = kPer / NR;         // columns a thread moves per tile
        int gcols[kCols], gn = 0;                // GATHER: the columns behind `pre`, and how many real ones the fetched tile holds
        TileCursor gnext = cur;                  // GATHER: the cursor after the fetched tile
        auto fetch_gathered = [&]() {
#pragma unroll
            for (int k = 0; k < kCols; ++k) {
                const u32x4* col = reinterpret_cast<const u32x4*>(pk.rec + ((long)b * M + max(gcols[k], 0)) * NR);
#pragma unroll
                for (int kb = 0; kb < NR; ++kb) pre[k * NR + kb] = col[kb];
            }
        };
        if (GATHER) {
            if (cur.q < q_end) {
                gnext = gather_tile<kCols, kThreads, kTile>(rg, M, q_end, split, ns, cur, tid, gcols, gn, pieces);
                fetch_gathered();
            }
        } else if (PRE && cur.q < q_end) {
            fetch_records<kPer, kThreads, !SPARSE, NR>(pre, tile_src(cur.j0), min(kTile, cur.je - cur.j0), tid);
        }

        while (cur.q < q_end) {
            {
                const int j0 = cur.j0;
                if (GATHER_NOW) gnext = gather_tile<kCols, kThreads, kTile>(rg, M, q_end, split, ns, cur, tid, gcols, gn, pieces);
                const int n = (GATHER || GATHER_NOW) ? gn : min(kTile, cur.je - j0);
                const int npad = (n + 31) & ~31;
                TileCursor nxt = cur;   // the tile after this one
                if (GATHER || GATHER_NOW) {
                    nxt = gnext;
                } else {
                    nxt.j0 += kTile;
                    if (nxt.j0 >= nxt.je) {
                        nxt.q += SPARSE ? ns : 1;
                        open_interval<SPARSE, PRE>(rg, M, q_end, split, ns, nxt);
                    }
                }
                __syncthreads();
                if (PRE && !SPARSE) {
#pragma unroll
                    for (int k = 0; k < kPer; ++k) {
                        const int r = tid + k * kThreads;
                        if (r < npad * NR) *reinterpret_cast<u32x4*>(&tileX[r]) = pre[k];
                    }
                } else if (PRE) {
#pragma unroll
                    for (int k = 0; k < kPer / NR; ++k) {
                        const int t = tid + k * kThreads;
                        if (t < npad) {
                            u32x4* dst = reinterpret_cast<u32x4*>(&tileX[(t >> 5) * GS + (t & 31)]);
                            const bool real = t < n;      // (gathered tiles fill their slots in order: the same test)
#pragma unroll
                            for (int kb = 0; kb < NR; ++kb) dst[kb * 32] = real ? pre[k * NR + kb] : neutral[kb];
                        }
                    }
                }
                if (GATHER) {
                    if (nxt.q < q_end) {
                        gnext = gather_tile<kCols, kThreads, kTile>(rg, M, q_end, split, ns, nxt, tid, gcols, gn, pieces);
                        fetch_gathered();
                    }
                } else if (PRE) {
                    if (nxt.q < q_end)
                        fetch_records<kPer, kThreads, !SPARSE, NR>(pre, tile_src(nxt.j0), min(kTile, nxt.je - nxt.j0), tid);
                } else if (GATHER_NOW) {
#pragma unroll
                    for (int k = 0; k < kCols; ++k) {
                        const int t = tid + k * kThreads;
                        if (t < npad) pack_column<D, T, L>(prm, (long)b * M + max(gcols[k], 0), t < n, centre, &tileX[(t >> 5) * GS + (t & 31)], 32);
                    }
                } else {
                    for (int t = tid; t < npad; t += kThreads)
                        pack_column<D, T, L>(prm, (long)b * M + j0 + t, t < n, centre, &tileX[(t >> 5) * GS + (t & 31)], 32);
                }
                cur = nxt;
                __syncthreads();
                if (!wave_active) continue;

                const int nG = npad / 32;
                int G0 = cpart;
                // one 32 x 32 block of exponents: column group G against row tile rt (`plain`: with n = 0, for the exact maxima)
                auto block = [&](int G, int rt, bool plain) -> f32x16 {
                    if constexpr (H2) {
                        const uint4 xa = plain ? select_u4(half == 0, xd_with_n<L>(Xlo[rt], 0.f), Xlo[rt]) : Xlo[rt];
                        return mfma_h32(tileX[G * GS + rec0], xa, zero16);
                    } else {
                        f32x16 u = mfma_x32(tileX[G * GS + rec0], Xlo[rt], zero16);
                        return mfma_x32(tileX[G * GS + 64 + rec0], select_u4(plain && half, kOnes, Xhi[rt]), u);
                    }
                };
                auto set_max = [&](int rt, float mx) {      // the running maximum enters through the K slots of the scalar item
                    if constexpr (H2) Xlo[rt] = select_u4(half == 0, xd_with_n<L>(Xlo[rt], -mx), Xlo[rt]);
                    else if (half) Xhi[rt] = pack_negmax(mx);
                };
                auto first_exact = [&](int G, int rt) {     // the first group a slot sees: exact maximum over its 32 columns (n = 0 so far)
                    const f32x16 u = block(G, rt, false);
                    float um = max16(u);
                    um = fmaxf(um, __shfl_xor(um, 32, 64));
                    um = fmaxf(um, kFloor);
                    m[rt] = um;
                    ssum[rt] = sum_exp2_16(u, um);
                    set_max(rt, um);
                    if (XS > 0) __builtin_amdgcn_sched_barrier(0);      // one slot at a time: 16 result registers, not 16 per slot
                };
                auto group_exact = [&](int G, int rt) {     // one group with its exact maximum folded into the running pair
                    const f32x16 u = block(G, rt, true);
                    float um = max16(u);
                    um = fmaxf(um, __shfl_xor(um, 32, 64));
                    const float mnew = fmaxf(m[rt], um);
                    ssum[rt] = ssum[rt] * fast_exp2(m[rt] - mnew) + sum_exp2_16(u, mnew);
                    m[rt] = mnew;
                    if (XS > 0) __builtin_amdgcn_sched_barrier(0);
                };
                if (first_group && G0 < nG) {   // exact maximum over the first 32 columns (of this wavefront's share)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) first_exact(G0, rt);
                    first_group = false;
                    G0 += cs;
                }
                int GX0 = wslot;                // leftover row tiles: this wavefront's column groups are wslot, wslot + 4, ...
                if (XS > 0 && nx > 0 && first_shared && GX0 < nG) {
#pragma unroll
                    for (int rt = RT; rt < NS; ++rt)
                        if (rt - RT < nx) first_exact(GX0, rt);
                    first_shared = false;
                    GX0 += 4;
                }

                float stmp[NS];
#pragma unroll
                for (int rt = 0; rt < NS; ++rt) stmp[rt] = 0.f;
                for (int G = G0; G < nG; G += cs) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) stmp[rt] += sum_exp2_16(block(G, rt, false));
                }
                if (XS > 0 && nx > 0 && !first_shared) {
                    for (int G = GX0; G < nG; G += 4) {
#pragma unroll
                        for (int rt = RT; rt < NS; ++rt)
                            if (rt - RT < nx) {
                                stmp[rt] += sum_exp2_16(block(G, rt, false));
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
                }
                float smax = stmp[0];
#pragma unroll
                for (int rt = 1; rt < NS; ++rt) smax = fmaxf(smax, stmp[rt]);
                if (__any(!(smax < kSumThr))) {
                    // a term far above the lazy max arrived (or inf / NaN): redo the tile with exact per-group maxima
                    for (int G = G0; G < nG; G += cs) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) group_exact(G, rt);
                    }
                    if (XS > 0 && nx > 0 && !first_shared) {
                        for (int G = GX0; G < nG; G += 4) {
#pragma unroll
                            for (int rt = RT; rt < NS; ++rt)
                                if (rt - RT < nx) group_exact(G, rt);
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < NS; ++rt)      // (a leftover slot that has seen no group yet keeps n = 0 for its first, exact one)
                        if (rt < RT || (rt - RT < nx && !first_shared)) set_max(rt, m[rt]);
                } else {
#pragma unroll
                    for (int rt = 0; rt < NS; ++rt) ssum[rt] += stmp[rt];
                }
            }
        }

        float sfin[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            sfin[rt] = ssum[rt] + __shfl_xor(ssum[rt], 32, 64);   // both halves carry the same max
            if (H2 && m[rt] <= kH2Floor * 0.98f) sfin[rt] = 0.f;  // a row still at the floor has seen no mass (glhip_softmin_xd.h)
        }
        if (cs > 1) {      // (workgroup-uniform) the wavefronts that shared a row tile merge their (max, sum) pairs through the tile buffer
            __syncthreads();                                      // everybody is done with the last tile
            float* red = reinterpret_cast<float*>(tileX);         // [wavefront slot][32 rows][max, sum]
            if (half == 0) {
                red[(wslot * 32 + l31) * 2] = m[0];
                red[(wslot * 32 + l31) * 2 + 1] = sfin[0];
            }
            __syncthreads();
            if (cpart == 0) {
                for (int c = 1; c < cs; ++c) {
                    const float m2 = red[((wslot + c) * 32 + l31) * 2], s2 = red[((wslot + c) * 32 + l31) * 2 + 1];
                    const float mnew = fmaxf(m[0], m2);
                    sfin[0] = sfin[0] * fast_exp2(m[0] - mnew) + s2 * fast_exp2(m2 - mnew);
                    m[0] = mnew;
                }
            }
            __syncthreads();                                      // (the next row pass stages into the same buffer)
        }

        auto write_row = [&](int i, float mrow, float s) {      // row i: running maximum of its exponents (without r_i) and their sum
            float xi[D];
            load_point<D, T>(prm.x, (long)b * N + i, xi);
            float n2 = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float xt = xi[d] - centre[d];
                n2 = __builtin_fmaf(xt, xt, n2);
            }
            const float mtot = __builtin_fmaf(-0.5f * prm.s2, n2, mrow);   // r_i + m
            if (ns == 1) {
                prm.out[(long)b * N + i] = finish_value(prm, (long)b * N + i, mtot + fast_log2(s));
            } else {
                float* dst = sp.workspace + split * sp.split_stride + ((long)b * N + i) * 2;
                dst[0] = mtot;
                dst[1] = s;
            }
        };
        if (wave_active && cpart == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int i = wave_row0 + rt * 32 + l31;
                if (half == 0 && i < row_end) write_row(i, m[rt], sfin[rt]);
            }
        }
        if (XS > 0 && nx > 0) {     // (workgroup-uniform) leftover row tiles: the four wavefronts' (max, sum) pairs meet in the tile buffer
            __syncthreads();                                      // everybody is done with the last tile
            float* red = reinterpret_cast<float*>(tileX);         // [leftover tile][wavefront slot][32 rows][max, sum]
#pragma unroll
            for (int rt = RT; rt < NS; ++rt) {
                if (rt - RT < nx) {
                    float sx = ssum[rt] + __shfl_xor(ssum[rt], 32, 64);      // (all 64 lanes take part in the exchange)
                    if (H2 && m[rt] <= kH2Floor * 0.98f) sx = 0.f;
                    if (half == 0) {
                        red[(((rt - RT) * 4 + wslot) * 32 + l31) * 2] = m[rt];
                        red[(((rt - RT) * 4 + wslot) * 32 + l31) * 2 + 1] = sx;
                    }
                }
            }
            __syncthreads();
            if (wslot < nx && half == 0) {                        // wavefront slot s finishes leftover tile s
                const float* mine = red + (wslot * 4 * 32 + l31) * 2;
                float mm = mine[0], sm = mine[1];
                for (int c = 1; c < 4; ++c) {
                    const float m2 = mine[c * 64], s2 = mine[c * 64 + 1];
                    const float mnew = fmaxf(mm, m2);
                    sm = sm * fast_exp2(mm - mnew) + s2 * fast_exp2(m2 - mnew);
                    mm = mnew;
                }
                const int i = xb + wslot * 32 + l31;
                if (i < xe) write_row(i, mm, sm);
            }
            __syncthreads();                                      // (a next row pass would stage into the same buffer)
        }
    }
}

// (the block-sparse 4-wavefront f16 x 2 kernel carries one more row-tile slot: held to 5 wavefronts per SIMD, <= 96 VGPRs)
template <int D, typename T, bool SPARSE, int RT, int NW, bool PRE = false, int L = XL_BF16X3>
__global__ void __launch_bounds__(NW * 64, (SPARSE && RT == 1 && NW == 4 && L == XL_F16X2) ? 5 : 1)
softmin_fwd_x32_kernel(SoftminParams<T> prm, Ranges rg, int N, int M, SplitInfo sp, PackedCols pk) {
    __shared__ uint4 tileX[fwd_tile(NW) * X32Layout<L>::NR];
    int bx, b, split;
    workgroup_coords(sp, bx, b, split);
    softmin_fwd_x32_body<D, T, SPARSE, RT, NW, PRE, L>(prm, rg, N, M, sp, pk, bx, b, split, tileX);
}

// Up to four independent dense reductions in ONE launch — the four soft-mins of a Sinkhorn iteration
// (glhip_sinkhorn_iter4).  grid = (max row blocks, B, n_splits * count); problem k = blockIdx.z / n_splits.
template <typename T>
struct SoftminMulti {
    SoftminParams<T> p[4];
    int N[4], M[4];
    long ws_stride;     // floats of split workspace per problem
    int count;
    PackedCols pk[4];   // PRE launches: the packed columns of every problem (pack_columns_multi_kernel)
};

// the columns of all the problems of a multi launch as packed records, once (grid: column blocks x batch x problem)
template <int D, typename T, int L = XL_BF16X3>
__global__ void __launch_bounds__(kBlock)
pack_columns_multi_kernel(SoftminMulti<T> m) {
    const int k = blockIdx.z, b = blockIdx.y;
    const int N = m.N[k], M = m.M[k];
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (N == 0 || j >= ((M + 31) & ~31)) return;
    float centre[D];
    launch_centre<D, T>(m.p[k].x, b, N, centre);
    pack_column<D, T, L>(m.p[k], (long)b * M + j, j < M, centre, m.pk[k].rec + b * m.pk[k].stride + (j >> 5) * (32 * X32Layout<L>::NR) + (j & 31), 32);
}

template <int D, typename T, int NW, bool PRE = false, int L = XL_BF16X3>
__global__ void __launch_bounds__(NW * 64)
softmin_fwd_x32_multi_kernel(SoftminMulti<T> m, SplitInfo sp) {
    __shared__ uint4 tileX[kTileX * X32Layout<L>::NR];
    const int k = blockIdx.z / sp.n_splits;
    const int split = blockIdx.z - k * sp.n_splits;
    const int N = m.N[k], M = m.M[k];
    if ((int)blockIdx.x * (NW * 32) >= N) return;
    SplitInfo spk = sp;
    spk.workspace += k * m.ws_stride;
    spk.split_stride = (long)gridDim.y * N * 2;   // this problem's own row count
    softmin_fwd_x32_body<D, T, false, 1, NW, PRE, L>(m.p[k], Ranges{nullptr, nullptr, nullptr}, N, M, spk, m.pk[k],
                                                  (int)blockIdx.x, (int)blockIdx.y, split, tileX);
}

template <class Op, typename T>
__global__ void __launch_bounds__(kBlock)
merge_multi_kernel(SoftminMulti<T> m, SplitInfo sp) {
    constexpr int D = Op::kDim;
    const int k = blockIdx.z;
    const int N = m.N[k];
    const int b = blockIdx.y;
    const int row0 = blockIdx.x * kBlock * Op::kRows;
    if (row0 >= N) return;
    float centre[D];
    Op::load_centre(m.p[k], b, N, row0, centre);
    for (int i = row0 + threadIdx.x; i < min(N, row0 + kBlock * Op::kRows); i += kBlock)
        Op::merge_row(m.p[k], b, N, i, centre, sp.workspace + k * m.ws_stride + ((long)b * N + i) * Op::kPartial, sp.n_splits,
                      (long)gridDim.y * N * Op::kPartial);
}

}  // namespace glhip
