// glhip_wsum_x32.h — the weighted-sum reductions of glhip_wsum_mfma.h on v_mfma_f32_32x32x16_bf16.
//
//     R_i[c] = sum_j  2^( [a_i,1].[yt_j,H_j] + C_i ) * q_j[c]          (modes: see WsumMode)
//
// Same bf16x3 exponent as the forward kernel (glhip_softmin_x32.h) and the same LDS record layout, but the block is
// NOT transposed: the MFMA rows are 32 rows x_i (registers), its columns 32 columns y_j (LDS records), so that lane l
// owns column j = l % 32 and the small per-column vector q_j is one register per component; its 16 result registers
// are 16 different rows (row 8*(v/4) + 4*(l/32) + v%4 of the 32-row block), each with its own accumulators.  The
// per-row constant C_i rides in the spare K slots of the x-side operand ([1,1,1,c1,c2,c3] against [H1,H2,H3,1,1,1]),
// the accumulator input is the inline constant 0.  Per 1024 pairs: 2 MFMA, 16 exp2, 16 x NQ fma (+16 adds for the
// plan-mass component of the soft-min gradient) — versus the 16x16x32 kernel it saves the ~6 issue cycles per MFMA
// (tools/ubench/overlap.hip) and, with pre-packed columns, the per-workgroup re-splitting of every tile.
#pragma once

#include "glhip_softmin_x32.h"
#include "glhip_wsum_mfma.h"

namespace glhip {

constexpr int kWsumNW = 8;                       // wavefronts per workgroup: 8 x 32 rows = kMfmaRowsPerBlock
static_assert(kWsumNW * 32 == kMfmaRowsPerBlock, "the merge kernels tile rows in blocks of kMfmaRowsPerBlock");

// q_j[c] and the H_j entering the exponent, for one column (coordinates relative to `centre`)
template <int MODE, int D, typename T>
__device__ __forceinline__ void wsum_column(const WsumParams<T>& prm, long col, bool valid, const float (&centre)[D],
                                            float (&rec)[4], float (&qv)[WsumShape<MODE, D>::kNQ]) {
    constexpr int NQ = WsumShape<MODE, D>::kNQ;
    rec[0] = rec[1] = rec[2] = 0.f;
    rec[3] = kNegBig;
#pragma unroll
    for (int c = 0; c < NQ; ++c) qv[c] = 0.f;
    if (valid) {
        float yj[D];
        load_point<D, T>(prm.y, col, yj);
        float n2 = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            rec[d] = yj[d] - centre[d];
            n2 = __builtin_fmaf(rec[d], rec[d], n2);
        }
        const float sj = prm.s[col];
        if (MODE == WS_SOFTMIN_BWD) {
            rec[3] = __builtin_fmaf(-0.5f * prm.s2, n2, sj * kLog2e);
#pragma unroll
            for (int d = 0; d < D; ++d) qv[d] = rec[d];
        } else {
            rec[3] = -0.5f * prm.s2 * n2;
            if (MODE == WS_GAUSS_FWD) qv[0] = sj;
            else {
#pragma unroll
                for (int d = 0; d < D; ++d) qv[d] = sj * rec[d];
                qv[D] = sj;
            }
        }
    }
}

struct PackedQ {
    float* q;       // [NQ][B*M]: component c of column (b, j) at q[c * total + b*M + j]
    long total;     // B*M
};

template <int MODE, int D, typename T, bool GROUPED>
__global__ void __launch_bounds__(kBlock)
wsum_pack_kernel(WsumParams<T> prm, int N, int M, PackedCols pk, PackedQ pq) {
    constexpr int NQ = WsumShape<MODE, D>::kNQ;
    const int b = blockIdx.y;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= (GROUPED ? ((M + 31) & ~31) : M)) return;
    float centre[D];
    launch_centre<D, T>(prm.x, b, N, centre);
    float rec[4], qv[NQ];
    wsum_column<MODE, D, T>(prm, (long)b * M + j, j < M, centre, rec, qv);
    uint4* base = GROUPED ? pk.rec + b * pk.stride + (j >> 5) * 128 + (j & 31) : pk.rec + ((long)b * M + j) * 4;
    const int stride = GROUPED ? 32 : 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) base[d * stride] = (d < D) ? pack_y(rec[d]) : uint4{0u, 0u, 0u, 0u};
    base[3 * stride] = pack_h1(rec[3]);
    if (j < M) {
#pragma unroll
        for (int c = 0; c < NQ; ++c) pq.q[c * pq.total + (long)b * M + j] = qv[c];
    }
}

template <int NQ>
__device__ __forceinline__ void fetch_q(float (&preq)[NQ], const PackedQ& pq, long col0, int n, int tid) {
#pragma unroll
    for (int c = 0; c < NQ; ++c) preq[c] = pq.q[c * pq.total + col0 + (tid < n ? tid : 0)];
}

template <int MODE, int D, typename T, bool SPARSE, bool PRE>
__global__ void __launch_bounds__(kWsumNW * 64)
wsum_x32_kernel(WsumParams<T> prm, Ranges rg, int N, int M, SplitInfo sp, PackedCols pk, PackedQ pq) {
    constexpr int NQ = WsumShape<MODE, D>::kNQ;
    constexpr int NA = WsumShape<MODE, D>::kNA;
    constexpr int kThreads = kWsumNW * 64;
    constexpr int kPer = (kTileX * 4) / kThreads;
    static_assert(kTileX == kThreads, "one column per thread and tile");
    __shared__ uint4 tileX[kTileX * 4];      // [column group of 32][K block][column], as in softmin_fwd_x32_kernel
    __shared__ float tileQ[NQ * kTileX];     // [component][column]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, b, split;
    workgroup_coords(sp, bx, b, split);
    const int ns = sp.n_splits;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int rec0 = half * 32 + l31;

    int row_begin, row_end, q_begin, q_end;
    block_extent<SPARSE>(rg, N, kMfmaRowsPerBlock, row_begin, row_end, q_begin, q_end, bx);

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int row0 = row_begin; row0 < row_end; row0 += kMfmaRowsPerBlock) {
        float centre[D];
        if (PRE) launch_centre<D, T>(prm.x, b, N, centre);
        else load_point<D, T>(prm.x, (long)b * N + row0, centre);
        const int wave_row0 = row0 + wave * 32;
        const bool wave_active = wave_row0 < row_end;

        // x-side operands of this lane's row l31 of the block: K blocks `half` and 2 + `half`
        uint4 Xlo, Xhi;
        {
            const int i = min(wave_row0 + l31, row_end - 1);
            float xi[D];
            load_point<D, T>(prm.x, (long)b * N + i, xi);
            float a[3] = {0.f, 0.f, 0.f};
            float n2 = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float xt = xi[d] - centre[d];
                a[d] = xt * prm.s2;
                n2 = __builtin_fmaf(xt, xt, n2);
            }
            float cst = -0.5f * prm.s2 * n2;                                                     // r_i
            if (MODE == WS_SOFTMIN_BWD) cst -= prm.fwd[(long)b * N + i] / prm.out_scale;         // -(LSE2_i - r_i)
            const uint4 z = uint4{0u, 0u, 0u, 0u};
            const uint4 p0 = pack_a(a[0]), p1 = (D > 1) ? pack_a(a[1]) : z, p2 = (D > 2) ? pack_a(a[2]) : z;
            Xlo = half ? p1 : p0;
            Xhi = half ? pack_negmax(-cst) : p2;   // [1,1,1,c1,c2,c3,0,0]
        }
        float acc[16][NA];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
#pragma unroll
            for (int c = 0; c < NA; ++c) acc[v][c] = 0.f;
        }

        u32x4 pre[kPer];
        float preq[NQ];
        const uint4 nh = pack_h1(kNegBig);
        const u32x4 neutral_h = u32x4{nh.x, nh.y, nh.z, nh.w};
        auto tile_src = [&](int j0) {
            return SPARSE ? pk.rec + ((long)b * M + j0) * 4 : pk.rec + b * pk.stride + (long)(j0 >> 5) * 128;
        };

        TileCursor cur;
        cur.q = q_begin + (SPARSE ? split : 0);
        cur.j0 = cur.je = 0;
        open_interval<SPARSE, PRE>(rg, M, q_end, split, ns, cur);
        if (PRE && cur.q < q_end) {
            fetch_records<kPer, kThreads, !SPARSE>(pre, tile_src(cur.j0), min(kTileX, cur.je - cur.j0), tid);
            fetch_q<NQ>(preq, pq, (long)b * M + cur.j0, min(kTileX, cur.je - cur.j0), tid);
        }

        while (cur.q < q_end) {
            const int j0 = cur.j0;
            const int n = min(kTileX, cur.je - j0);
            const int npad = (n + 31) & ~31;
            TileCursor nxt = cur;
            nxt.j0 += kTileX;
            if (nxt.j0 >= nxt.je) {
                nxt.q += SPARSE ? ns : 1;
                open_interval<SPARSE, PRE>(rg, M, q_end, split, ns, nxt);
            }
            __syncthreads();
            if (PRE && !SPARSE) {
#pragma unroll
                for (int k = 0; k < kPer; ++k) {
                    const int r = tid + k * kThreads;
                    if (r < npad * 4) *reinterpret_cast<u32x4*>(&tileX[r]) = pre[k];
                }
            } else if (PRE) {
                if (tid < npad) {
                    u32x4* dst = reinterpret_cast<u32x4*>(&tileX[(tid >> 5) * 128 + (tid & 31)]);
                    const bool real = tid < n;
#pragma unroll
                    for (int kb = 0; kb < 3; ++kb) dst[kb * 32] = real ? pre[kb] : u32x4{0u, 0u, 0u, 0u};
                    dst[96] = real ? pre[3] : neutral_h;
                }
            }
            if (PRE) {
                if (tid < npad) {
#pragma unroll
                    for (int c = 0; c < NQ; ++c) tileQ[c * kTileX + tid] = (tid < n) ? preq[c] : 0.f;
                }
                if (nxt.q < q_end) {
                    fetch_records<kPer, kThreads, !SPARSE>(pre, tile_src(nxt.j0), min(kTileX, nxt.je - nxt.j0), tid);
                    fetch_q<NQ>(preq, pq, (long)b * M + nxt.j0, min(kTileX, nxt.je - nxt.j0), tid);
                }
            } else if (tid < npad) {
                float rec[4], qv[NQ];
                wsum_column<MODE, D, T>(prm, (long)b * M + j0 + tid, tid < n, centre, rec, qv);
                uint4* base = &tileX[(tid >> 5) * 128 + (tid & 31)];
#pragma unroll
                for (int d = 0; d < 3; ++d) base[d * 32] = (d < D) ? pack_y(rec[d]) : uint4{0u, 0u, 0u, 0u};
                base[96] = pack_h1(rec[3]);
#pragma unroll
                for (int c = 0; c < NQ; ++c) tileQ[c * kTileX + tid] = qv[c];
            }
            cur = nxt;
            __syncthreads();
            if (!wave_active) continue;

            for (int G = 0; G < npad / 32; ++G) {
                const uint4 ya = tileX[G * 128 + rec0], yb = tileX[G * 128 + 64 + rec0];
                float Q[NQ];
#pragma unroll
                for (int c = 0; c < NQ; ++c) Q[c] = tileQ[c * kTileX + G * 32 + l31];
                f32x16 d = mfma_x32(Xlo, ya, zero16);
                d = mfma_x32(Xhi, yb, d);
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float w = fast_exp2(d[v]);
#pragma unroll
                    for (int c = 0; c < NQ; ++c) acc[v][c] = __builtin_fmaf(w, Q[c], acc[v][c]);
                    if (MODE == WS_SOFTMIN_BWD) acc[v][D] += w;
                }
            }
        }

        if (wave_active) {
            // sum over the 32 column-lanes of each half: afterwards every lane holds the totals of its half's 16 rows
#pragma unroll
            for (int v = 0; v < 16; ++v) {
#pragma unroll
                for (int c = 0; c < NA; ++c) {
#pragma unroll
                    for (int off = 1; off < 32; off <<= 1) acc[v][c] += __shfl_xor(acc[v][c], off, 64);
                }
            }
            if (l31 == 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int i = wave_row0 + 8 * (v >> 2) + 4 * half + (v & 3);
                    if (i < row_end) {
                        float a_[NA];
#pragma unroll
                        for (int c = 0; c < NA; ++c) a_[c] = acc[v][c];
                        float xt[D];
                        {
                            float xi[D];
                            load_point<D, T>(prm.x, (long)b * N + i, xi);
#pragma unroll
                            for (int d = 0; d < D; ++d) xt[d] = xi[d] - centre[d];
                        }
                        float* part = sp.workspace + split * sp.split_stride + ((long)b * N + i) * WsumShape<MODE, D>::kPart;
                        if (MODE == WS_SOFTMIN_BWD) {
                            if (ns == 1) {
                                const float gi = prm.g[(long)b * N + i];
                                const float inv = (a_[D] > 0.f) ? 1.0f / a_[D] : 0.f;
#pragma unroll
                                for (int d = 0; d < D; ++d) prm.gx[((long)b * N + i) * D + d] = gi * (xt[d] - a_[d] * inv);
                            } else {
                                // the merge (SoftminBwdOp::merge_row) works relative to the first row of the 256-row pass
                                float cm[D];
                                load_point<D, T>(prm.x, (long)b * N + row0, cm);
#pragma unroll
                                for (int d = 0; d < D; ++d) part[d] = a_[d] + a_[D] * (centre[d] - cm[d]);
                                part[D] = a_[D];
                            }
                        } else if (MODE == WS_GAUSS_FWD) {
                            if (ns == 1) prm.out[(long)b * N + i] = a_[0];
                            else part[0] = a_[0];
                        } else {
                            if (ns == 1) {
                                const float gi = prm.g[(long)b * N + i] * prm.gscale;
#pragma unroll
                                for (int d = 0; d < D; ++d) prm.gx[((long)b * N + i) * D + d] = gi * (xt[d] * a_[D] - a_[d]);
                            } else {
#pragma unroll
                                for (int d = 0; d < D; ++d) part[d] = prm.tscale * (xt[d] * a_[D] - a_[d]);
                            }
                        }
                    }
                }
            }
        }
    }
}

}  // namespace glhip
