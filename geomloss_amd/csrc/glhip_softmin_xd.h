// glhip_softmin_xd.h — the bf16x3 matrix-core reductions for clouds of dimension 4 <= D <= 16: soft-min forward (p = 2, incl. the
// fused Sinkhorn half-step) and gaussian kernel product.
//
// Same transposed 32 x 32 blocks as glhip_softmin_x32.h (D <= 3) and the same idea — every fp32 operand is the exact sum of three
// bf16 pieces and an exponent is a short dot product on v_mfma_f32_32x32x16_bf16 — with a denser K layout (round 4).  A coordinate
// takes SIX slots, the products a1 y1, a1 y2, a2 y1, a1 y3, a3 y1, a2 y2 ([y1,y2,y1,y3,y1,y2] against [a1,a1,a2,a1,a3,a2]); the
// two products of relative size 2^-24 that the 8-slot layout of the D <= 3 kernels also carries (a2 y3, a3 y2) are dropped.  The
// pieces are split with ROUND-TO-NEAREST here (signed residuals half as large as truncated ones: |a2| <= 2^-9 |a|, |a3| <= 2^-17
// |a|), so the dropped terms are <= 2^-25 |a y| each, of either sign: below the 2^-24 rounding of a float32 product, and of the
// float32 accumulation inside the MFMA.  One more item of six slots carries the per-column scalar and the per-row constant
// ([H1,H2,H3,1,1,1] against [1,1,1,n1,n2,n3]); it comes FIRST, so that it always lies in the first 16-byte record of lane half 0.
// The items are laid end to end over the K dimension: 6 (D + 1) slots = NM = ceil(6 (D + 1) / 16) chained MFMAs —
//     D      4  5  6  7  8  9  10  11  12  13  14  15  16
//     NM     2  3  3  3  4  4   5   5   5   6   6   6   7        (8-slot layout of round 3: 3 3 4 4 5 5 6 6 7 7 8 8 9)
// Why the count matters (tools/ubench/chain.hip, profiles/r04_ubench_chain.txt, r04_xd_pmc.txt): next to the 16 v_exp_f32 +
// 16 v_add_f32 of a block every MFMA costs ~13 issue cycles while the VALU is the bound (200 + 13 NM cycles per 1024 pairs up to
// NM ~ 5) and its full 32-36 cycles beyond; the two pipes do overlap across the wavefronts of a SIMD, an in-wave software pipeline
// adds nothing from 2 waves per SIMD on; and the chip clocks down with the matrix load (2.22 GHz at D = 3, 1.94 at NM = 5,
// 1.74 at NM = 9).  Fewer MFMAs is the one lever that pays twice.
//
// The reference takes any D (`Vi({D})` in lse_genred, _legacy/sinkhorn_samples.py:322-334; its multiscale tutorial is 4-D:
// examples/sinkhorn_multiscale/plot_optimal_transport_cluster.py:58-61); before this header every D >= 4 went to the
// one-thread-per-row VALU kernel of glhip_generic.h (2 instructions per pair and dimension), which stays for D > 16, p = 1 and
// the gradients.
//
// LDS: NBP = 2 NM records of 16 bytes per column (the last one is a zero pad when D + 1 is odd).
//   * on-the-fly staging (small and block-sparse launches): tiles of 512 / 256 / 128 columns (<= 48 KiB); every workgroup splits
//     the columns of a tile into bf16 pieces itself (~13 VALU instructions per coordinate and column, shared by the RT x NW x 32
//     rows of its pass: 11 % of the VALU work of a pass at D = 8 with 512 rows, 43 % at D = 16 with 256);
//   * pre-packed columns (PRE, big dense launches; round 4): `xd_pack_kernel` writes the records of all columns ONCE, in the LDS
//     tile order, and the reducing workgroups copy whole tiles with LDS-DMA (`global_load_lds_dwordx4`: 1 KiB per wavefront
//     instruction, no staging registers, no VALU) into one of TWO tile buffers while the other one is being consumed: one
//     barrier per tile instead of two, and the matrix / VALU pipes never wait for the packing.  One centre per launch
//     (launch_centre), as for the D <= 3 kernel.
#pragma once

#include "glhip_klayout.h"
#include "glhip_softmin_x32.h"

namespace glhip {

enum XdMode { XD_SOFTMIN = 0, XD_GAUSS = 1 };

// One column as NBP records `stride` apart starting at `base`; coordinates relative to `centre`.
//   XD_SOFTMIN: H = log2(e) h_j - s/2 |yt|^2  (h_j through dual_entry: the fused half-step adds pot_j / eps)
//   XD_GAUSS:   H = -s/2 |yt|^2, and the weight v_j = prm.h[col] goes to *vdst
template <int MODE, int D, typename T, int L = XL_BF16X3>
__device__ __forceinline__ void pack_column_xd(const SoftminParams<T>& prm, long col, bool valid, const float (&centre)[D],
                                               uint4* base, int stride, float* vdst) {
    using S = XdShape<D, L>;
    float yt[D];
    float H = kNegBig, vj = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) yt[d] = 0.f;
    if (valid) {
        float yj[D];
        load_point<D, T>(prm.y, col, yj);
        float n2 = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            yt[d] = yj[d] - centre[d];
            n2 = __builtin_fmaf(yt[d], yt[d], n2);
        }
        if (MODE == XD_SOFTMIN) {
            H = __builtin_fmaf(-0.5f * prm.s2, n2, dual_entry(prm, col) * kLog2e);
        } else {
            H = -0.5f * prm.s2 * n2;
            vj = prm.h[col];
        }
    }
    if constexpr (L == XL_F16X2) {      // both sides carry sqrt(s); exponents below the floor are the floor
        const float q = __builtin_sqrtf(prm.s2);
#pragma unroll
        for (int d = 0; d < D; ++d) yt[d] *= q;
        H = __builtin_fmaxf(H, kH2Floor);
    }
#pragma unroll
    for (int r = 0; r < S::NBP; ++r) base[r * stride] = xd_record_of<D, false, L>(r, H, yt);      // record by record: few live pieces
    if (MODE == XD_GAUSS) *vdst = vj;
}

// Packed records of a whole launch: [B][ceil(M / 32)][NBP K blocks][32 columns] — the LDS tile layout, so that a tile (which
// starts on a group boundary) is a contiguous run.  Columns j >= M of the last group are neutral (H = -big, zero coordinates).
struct XdPacked {
    uint4* rec;
    long stride;    // records per batch item = ceil(M / 32) * 32 * NBP
};

template <int MODE, int D, typename T, int L = XL_BF16X3>
__global__ void __launch_bounds__(kBlock)
xd_pack_kernel(SoftminParams<T> prm, int N, int M, XdPacked pk) {
    using S = XdShape<D, L>;
    const int b = blockIdx.y;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= ((M + 31) & ~31)) return;
    float centre[D];
    launch_centre<D, T>(prm.x, b, N, centre);
    float unused;
    pack_column_xd<MODE, D, T, L>(prm, (long)b * M + j, j < M, centre, pk.rec + b * pk.stride + (long)(j >> 5) * S::kGroupRecs + (j & 31), 32,
                                  &unused);
}

// LDS-DMA: lane l of the wavefront copies 16 (4) bytes from its own global address to lds_base + 16 l (4 l); lds_base must be
// wave-uniform.  Completion is tracked by vmcnt of the issuing wavefront; readers need that wait followed by a barrier.
__device__ __forceinline__ void glds16(const uint4* gsrc, uint4* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* gsrc, float* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}

// the chained MFMAs of one 32 x 32 block: column group `g` (LDS) against the x-side operands X
template <int NM, int NBP, int L = XL_BF16X3>
__device__ __forceinline__ f32x16 xd_block(const uint4* __restrict__ g, int rec0, const uint4 (&X)[NM], const f32x16& zero16) {
    f32x16 u = (L == XL_F16X2) ? mfma_h32(g[rec0], X[0], zero16) : mfma_x32(g[rec0], X[0], zero16);
#pragma unroll
    for (int m = 1; m < NM; ++m) u = (L == XL_F16X2) ? mfma_h32(g[m * 64 + rec0], X[m], u) : mfma_x32(g[m * 64 + rec0], X[m], u);
    return u;
}

// sum_k 2^(u_k) v_k over the 16 registers of a block; register k <-> column (k / 4) * 8 + 4 * half + k % 4 of the group
// (vg = &tileV[first column of the group + 4 * half]: four broadcast 16-byte reads)
__device__ __forceinline__ float xd_weighted_sum(const f32x16& u, const float* __restrict__ vg) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v4 = *reinterpret_cast<const float4*>(vg + q * 8);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(fast_exp2(u[q * 4 + r]), vv[r], acc[r]);
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// LDS of one workgroup: tile buffers (two with pre-packed columns) and, for the gaussian product, the weights of the tile
template <int MODE, int D, bool PRE, int L>
struct XdLds {
    using S = XdShape<D, L>;
    static constexpr int kTileD = PRE ? S::kTilePre : S::kTile;
    static constexpr int kBufs = PRE ? 2 : 1;
    static constexpr int kTileV = (kTileD + 63) & ~63;               // weights travel 64 at a time (one 4-byte LDS-DMA instruction)
    static constexpr int kRecs = kBufs * kTileD * S::NBP;            // [buffer][column group of 32][K block 0..NBP-1][column]
    static constexpr int kWeights = MODE == XD_GAUSS ? kBufs * kTileV : 4;
};

// The work of one workgroup: row block bx of batch item b, column split `split`.
template <int MODE, int D, typename T, bool SPARSE, int RT, int NW, bool PRE, int L>
__device__ __forceinline__ void xd_fwd_body(const SoftminParams<T>& prm, const Ranges& rg, int N, int M, const SplitInfo& sp, const XdPacked& pk,
                                            int bx, int b, int split, uint4* tileBuf, float* tileVBuf) {
    using S = XdShape<D, L>;
    constexpr float kFloor = (L == XL_F16X2) ? kH2Floor : kMinusHuge;      // the running maximum of a row that has seen no mass yet
    static_assert(!(PRE && SPARSE), "pre-packed columns serve dense launches");
    constexpr int NM = S::NM, NBP = S::NBP, kTileD = PRE ? S::kTilePre : S::kTile;
    constexpr int kRowsPerWave = RT * 32;
    constexpr int kRowsPerBlock = NW * kRowsPerWave;
    constexpr int kThreads = NW * 64;
    constexpr int kTileV = XdLds<MODE, D, PRE, L>::kTileV;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ns = sp.n_splits;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int rec0 = half * 32 + l31;       // K block `half` of MFMA 0; MFMA m: + 64 m

    int row_begin, row_end, q_begin, q_end;
    block_extent<SPARSE>(rg, N, kRowsPerBlock, row_begin, row_end, q_begin, q_end, bx);

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool owns_h = (half == 0);        // the scalar item (slots 0..5) lies in record 0: lane half 0 of the first MFMA

    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerBlock) {
        float centre[D];
        if (PRE) launch_centre<D, T>(prm.x, b, N, centre);
        else load_point<D, T>(prm.x, (long)b * N + row0, centre);

        const int wave_row0 = row0 + wave * kRowsPerWave;
        const bool wave_active = wave_row0 < row_end;
        uint4 X[RT][NM];
        float m[RT], ssum[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int i = min(wave_row0 + rt * 32 + l31, row_end - 1);
            float xi[D];
            load_point<D, T>(prm.x, (long)b * N + i, xi);
            float a[D], n2 = 0.f;
            const float xscale = (L == XL_F16X2) ? __builtin_sqrtf(prm.s2) : prm.s2;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float xt = xi[d] - centre[d];
                n2 = __builtin_fmaf(xt, xt, n2);
                a[d] = xt * xscale;
            }
            // scalar item of the x side: [1,1,1,n1,n2,n3]; soft-min: n = -running max (0 until the first group has been seen),
            // gaussian: n = r_i = -s/2 |xt_i|^2, constant
            float nrow = (MODE == XD_SOFTMIN) ? 0.f : -0.5f * prm.s2 * n2;
            if (L == XL_F16X2) nrow = __builtin_fmaxf(nrow, kH2Floor);
#pragma unroll
            for (int mm = 0; mm < NM; ++mm)     // this lane's half of MFMA mm: record 2 mm + half
                X[rt][mm] = select_u4(half != 0, xd_record_of<D, true, L>(2 * mm + 1, nrow, a), xd_record_of<D, true, L>(2 * mm, nrow, a));
            m[rt] = kFloor;
            ssum[rt] = 0.f;
        }
        bool first_group = (MODE == XD_SOFTMIN);

        // PRE: tile t + 1 of the split travels into the other buffer (LDS-DMA, issued right after the barrier that ends tile t - 1)
        // while tile t is consumed.  The split's columns are whole groups of 32, like the packed layout.
        int pre_js = 0, pre_je = 0;
        if (PRE) {
            const int len = (((M + ns - 1) / ns) + 31) & ~31;
            pre_js = min(M, split * len);
            pre_je = min(M, pre_js + len);
        }
        auto stage = [&](int j0, int buf) {
            const int nGs = (min(kTileD, pre_je - j0) + 31) >> 5;
            const uint4* src = pk.rec + b * pk.stride + (long)(j0 >> 5) * S::kGroupRecs;
            for (int c = wave; c < nGs * NM; c += NW) glds16(src + c * 64 + lane, &tileBuf[buf * (kTileD * NBP) + c * 64]);
            if (MODE == XD_GAUSS) {      // 64 weights per wavefront instruction; columns past M repeat the last one (their H is -big)
                for (int c = wave; c < (nGs + 1) / 2; c += NW)
                    glds4(prm.h + (long)b * M + min(j0 + c * 64 + lane, M - 1), &tileVBuf[buf * kTileV + c * 64]);
            }
        };
        if (PRE && pre_js < pre_je) stage(pre_js, 0);
        int pre_t = 0;

        // one tile: n real columns (padded to whole groups) starting at column j0 of the interval ending at je — or, block-sparse,
        // the gathered columns gcols (glhip_softmin_x32.h: gather_tile)
        constexpr int kCols = (kTileD + kThreads - 1) / kThreads;
        auto tile_body = [&](int j0, int je, int n, const int (&gcols)[kCols]) {
                const int npad = (n + 31) & ~31;
                const uint4* tile = tileBuf;
                const float* tileV = tileVBuf;
                if (PRE) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's share of tile t has landed ...
                    __syncthreads();                                      // ... and so has everybody's; tile t - 1 is consumed
                    if (j0 + kTileD < je) stage(j0 + kTileD, (pre_t + 1) & 1);
                    tile = tileBuf + (pre_t & 1) * (kTileD * NBP);
                    tileV = tileVBuf + (MODE == XD_GAUSS ? (pre_t & 1) * kTileV : 0);
                    ++pre_t;
                } else {
                    __syncthreads();
                    if (SPARSE) {      // gathered tile: slot t holds column gcols[k] (t = tid + k kThreads)
#pragma unroll
                        for (int k = 0; k < kCols; ++k) {
                            const int t = tid + k * kThreads;
                            if (t < npad)
                                pack_column_xd<MODE, D, T, L>(prm, (long)b * M + max(gcols[k], 0), t < n, centre,
                                                           &tileBuf[(t >> 5) * (32 * NBP) + (t & 31)], 32, &tileVBuf[MODE == XD_GAUSS ? t : 0]);
                        }
                    } else {
                        for (int t = tid; t < npad; t += kThreads)
                            pack_column_xd<MODE, D, T, L>(prm, (long)b * M + j0 + t, t < n, centre, &tileBuf[(t >> 5) * (32 * NBP) + (t & 31)], 32,
                                                       &tileVBuf[MODE == XD_GAUSS ? t : 0]);
                    }
                    __syncthreads();
                }
                if (!wave_active) return;

                const int nG = npad / 32;
                if (MODE == XD_GAUSS) {
                    for (int G = 0; G < nG; ++G) {
                        const uint4* g = &tile[G * (32 * NBP)];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            ssum[rt] += xd_weighted_sum(xd_block<NM, NBP, L>(g, rec0, X[rt], zero16), &tileV[G * 32 + half * 4]);
                    }
                    return;
                }

                int G0 = 0;
                if (first_group) {   // exact maximum over the first 32 columns
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x16 u = xd_block<NM, NBP, L>(&tile[0], rec0, X[rt], zero16);
                        float um = max16(u);
                        um = fmaxf(um, __shfl_xor(um, 32, 64));
                        um = fmaxf(um, kFloor);
                        m[rt] = um;
                        ssum[rt] = sum_exp2_16(u, um);
                        X[rt][0] = select_u4(owns_h, xd_with_n<L>(X[rt][0], -um), X[rt][0]);
                    }
                    first_group = false;
                    G0 = 1;
                }

                float stmp[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) stmp[rt] = 0.f;
                for (int G = G0; G < nG; ++G) {
                    const uint4* g = &tile[G * (32 * NBP)];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) stmp[rt] += sum_exp2_16(xd_block<NM, NBP, L>(g, rec0, X[rt], zero16));
                }
                float smax = stmp[0];
#pragma unroll
                for (int rt = 1; rt < RT; ++rt) smax = fmaxf(smax, stmp[rt]);
                if (__any(!(smax < kSumThr))) {
                    // a term far above the lazy max arrived (or inf / NaN): redo the tile with exact per-group maxima
                    for (int G = G0; G < nG; ++G) {
                        const uint4* g = &tile[G * (32 * NBP)];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            uint4 Xp[NM];
#pragma unroll
                            for (int mm = 0; mm < NM; ++mm) Xp[mm] = X[rt][mm];
                            Xp[0] = select_u4(owns_h, xd_with_n<L>(Xp[0], 0.f), Xp[0]);      // n = 0: plain exponents
                            const f32x16 u = xd_block<NM, NBP, L>(g, rec0, Xp, zero16);
                            float um = max16(u);
                            um = fmaxf(um, __shfl_xor(um, 32, 64));
                            const float mnew = fmaxf(m[rt], um);
                            ssum[rt] = ssum[rt] * fast_exp2(m[rt] - mnew) + sum_exp2_16(u, mnew);
                            m[rt] = mnew;
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) X[rt][0] = select_u4(owns_h, xd_with_n<L>(X[rt][0], -m[rt]), X[rt][0]);
                } else {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) ssum[rt] += stmp[rt];
                }
        };
        if (SPARSE) {
            // Block-sparse launches walk the concatenation of their column intervals: with small clusters (SplitInfo::gather) a tile
            // gathers as many intervals as fit, otherwise one piece of one interval, as before
            TileCursor cur;
            cur.q = q_begin + split;
            cur.j0 = cur.je = 0;
            open_interval<true, true>(rg, M, q_end, split, ns, cur);
            const int pieces = sp.gather ? kTileD : 1;
            while (cur.q < q_end) {
                int gcols[kCols], n = 0;
                const TileCursor nxt = gather_tile<kCols, kThreads, kTileD>(rg, M, q_end, split, ns, cur, tid, gcols, n, pieces);
                tile_body(0, 0, n, gcols);
                cur = nxt;
            }
        } else {
            int js, je;
            if (PRE) { js = pre_js; je = pre_je; }
            else column_interval<false>(rg, M, 0, split, ns, js, je);
            const int none[kCols] = {};
            for (int j0 = js; j0 < je; j0 += kTileD) tile_body(j0, je, min(kTileD, je - j0), none);
        }

        if (wave_active) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float s = ssum[rt] + __shfl_xor(ssum[rt], 32, 64);   // the halves hold the two 16-column halves of every block
                // XL_F16X2: a row still at the floor has seen padded / massless columns only, whose exponents were the floor too
                // (2^0 each): its sum is empty, as in the bf16 layout where such exponents are -1e30 / -inf
                if (L == XL_F16X2 && MODE == XD_SOFTMIN && m[rt] <= kH2Floor * 0.98f) s = 0.f;
                const int i = wave_row0 + rt * 32 + l31;
                if (half == 0 && i < row_end) {
                    const long idx = (long)b * N + i;
                    if (MODE == XD_GAUSS) {
                        if (ns == 1) prm.out[idx] = s;
                        else sp.workspace[split * sp.split_stride + idx] = s;
                    } else {
                        float xi[D];
                        load_point<D, T>(prm.x, idx, xi);
                        float n2 = 0.f;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            const float xt = xi[d] - centre[d];
                            n2 = __builtin_fmaf(xt, xt, n2);
                        }
                        const float mtot = __builtin_fmaf(-0.5f * prm.s2, n2, m[rt]);   // r_i + m
                        if (ns == 1) {
                            prm.out[idx] = finish_value(prm, idx, mtot + fast_log2(s));
                        } else {
                            float* dst = sp.workspace + split * sp.split_stride + idx * 2;
                            dst[0] = mtot;
                            dst[1] = s;
                        }
                    }
                }
            }
        }
    }
}

template <int MODE, int D, typename T, bool SPARSE, int RT, int NW, bool PRE, int L = XL_BF16X3>
__global__ void __launch_bounds__(NW * 64)
xd_fwd_kernel(SoftminParams<T> prm, Ranges rg, int N, int M, SplitInfo sp, XdPacked pk) {
    using LDS = XdLds<MODE, D, PRE, L>;
    __shared__ uint4 tileBuf[LDS::kRecs];
    __shared__ __attribute__((aligned(16))) float tileVBuf[LDS::kWeights];    // gaussian: the weights v_j of the tile (read back as float4)
    int bx, b, split;
    workgroup_coords(sp, bx, b, split);
    xd_fwd_body<MODE, D, T, SPARSE, RT, NW, PRE, L>(prm, rg, N, M, sp, pk, bx, b, split, tileBuf, tileVBuf);
}

// Up to four independent dense soft-min reductions in ONE launch — the four updates of a Sinkhorn iteration (glhip_sinkhorn_iter4)
// for clouds of dimension 4 <= D <= 16 (round 5): the multi launch of glhip_softmin_x32.h on this kernel's body; columns packed on the
// fly.  grid = (max row blocks, B, n_splits * count); problem k = blockIdx.z / n_splits.
template <int D, typename T, int NW, int L = XL_BF16X3>
__global__ void __launch_bounds__(NW * 64)
xd_fwd_multi_kernel(SoftminMulti<T> m, SplitInfo sp) {
    using LDS = XdLds<XD_SOFTMIN, D, false, L>;
    __shared__ uint4 tileBuf[LDS::kRecs];
    __shared__ __attribute__((aligned(16))) float tileVBuf[LDS::kWeights];
    const int k = blockIdx.z / sp.n_splits;
    const int split = blockIdx.z - k * sp.n_splits;
    const int N = m.N[k], M = m.M[k];
    if ((int)blockIdx.x * (NW * 32) >= N) return;
    SplitInfo spk = sp;
    spk.workspace += k * m.ws_stride;
    spk.split_stride = (long)gridDim.y * N * 2;   // this problem's own row count
    xd_fwd_body<XD_SOFTMIN, D, T, false, 1, NW, false, L>(m.p[k], Ranges{nullptr, nullptr, nullptr, nullptr}, N, M, spk, XdPacked{nullptr, 0},
                                                          (int)blockIdx.x, (int)blockIdx.y, split, tileBuf, tileVBuf);
}

}  // namespace glhip
