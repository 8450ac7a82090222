// glhip_wsum_mfma.h — matrix-core kernels for the reductions of the form
//     R_i[c] = sum_j  2^( [a_i,1].[yt_j,H_j] + C_i ) * q_j[c]
// i.e. "weights from a K = 4 fp32 MFMA + exp2, times a small per-column vector".  Three users:
//   WS_SOFTMIN_BWD : gradient of the p = 2 soft-min.  C_i = -(LSE2_i - r_i) from the saved forward, so the
//                    weights are the transport plan row P_ij <= 1;  q_j = (yt_j, 1).
//   WS_GAUSS_FWD   : gaussian kernel product.  C_i = r_i = -s/2 |xt_i|^2, H_j = -s/2 |yt_j|^2, s = log2(e)/blur^2,
//                    weights k_ij <= 1;  q_j = v_j.
//   WS_GAUSS_BWD   : its gradient in x.  q_j = (v_j yt_j, v_j);  grad = -(g_i/blur^2) (xt_i S0 - S1).
//   WS_GAUSS_FWDGRAD : product AND its row gradient in one pass — S0 is the product itself: out_i = S0,
//                    dout_i/dx_i = -(1/blur^2) (xt_i S0 - S1).  Used by the autograd forward when x requires gradients, so that
//                    the backward pass of a kernel norm is elementwise (fwd + bwd of the gaussian MMD: 3 + 2 reductions -> 3).
// Exponents are <= 0 by construction, so there is no running max.  Same wave / LDS layout and the same
// bf16 x 3 exponent MFMA as the forward soft-min (glhip_softmin_xdl.h): lane l holds D rows 4*(l/16)+r and
// column l%16 of each 16-column group.
// q is staged as tileQ[c][G][j] = float4 over the 4 column groups of super-group G: the 4 lanes that share a
// column read the same 16 bytes (broadcast), the 16 columns are contiguous (conflict-free).
//
// Accuracy note: like the expanded soft-min, each weight carries a relative error ~ 2^-23 * (s |xt|^2 + s |yt|^2);
// rows are re-centred per workgroup, so for cluster-sorted clouds (multiscale) this is ~1e-6, for unsorted
// clouds up to ~1e-4 per term (random sign).  The direct-difference VALU operators remain available
// (GLHIP_FLAG_NO_MFMA) and are what laplacian / energy / p = 1 always use.
#pragma once

#include "glhip_softmin_xdl.h"

namespace glhip {

enum WsumMode { WS_SOFTMIN_BWD = 0, WS_GAUSS_FWD = 1, WS_GAUSS_BWD = 2, WS_GAUSS_FWDGRAD = 3 };

template <typename T>
struct WsumParams {
    const T* x;         // (B,N,D)
    const T* y;         // (B,M,D)
    const float* s;     // (B,M): h (soft-min) or v (kernel product)
    const float* fwd;   // WS_SOFTMIN_BWD: saved forward (B,N)
    const float* g;     // bwd modes: (B,N)
    float* out;         // WS_GAUSS_FWD: (B,N)
    float* gx;          // bwd modes: (B,N,D)
    float s2;           // log2(e)/eps  or  log2(e)/blur^2
    float out_scale;    // soft-min: -eps ln2
    float gscale;       // WS_GAUSS_BWD: -1/blur^2
    float tscale;       // WS_GAUSS_BWD: coordinate pre-scale of the VALU operator (partials are stored in its units)
};

// acc[r] += w[r] * q.  NOTE (measured on MI355X, ROCm 7.2): when this was emitted as v_pk_fma_f32 and the
// next instruction was an XDL MFMA whose vdst reused the pk op's source registers (WAR), the low result of
// the pk op was sporadically lost in lanes 48-63 (one row of the gaussian gradient off by exactly one column
// group's contribution, in ~9 of 10 processes).  hipcc pads no hazard there.  The library is therefore built
// with packed fp32 math disabled (-target-feature -packed-fp32-ops, see csrc/Makefile); cost: ~5 % on the
// forward kernel, nothing elsewhere.
__device__ __forceinline__ void fma4(f32x4& acc, const f32x4& w, float q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(w[r], q, acc[r]);
}

template <int MODE, int D> struct WsumShape {
    static constexpr int kNQ = (MODE == WS_SOFTMIN_BWD) ? D : (MODE == WS_GAUSS_FWD ? 1 : D + 1);   // LDS q vectors
    static constexpr int kNA = (MODE == WS_GAUSS_FWD) ? 1 : D + 1;                                  // accumulators
    // floats per row in the split workspace = kPartial of the VALU operator whose merge_row finishes the job
    static constexpr int kPart = (MODE == WS_SOFTMIN_BWD || MODE == WS_GAUSS_FWDGRAD) ? D + 1 : (MODE == WS_GAUSS_FWD ? 1 : D);
};

// RT 16-row tiles per wavefront x NW wavefronts = 256 rows per workgroup in both shapes: (4, 4), or (2, 8) — half the accumulators
// per lane, twice the wavefronts
template <int MODE, int D, typename T, bool SPARSE, int RT = kMfmaRT, int NW = 4>
__global__ void __launch_bounds__(NW * 64)
wsum_mfma_kernel(WsumParams<T> prm, Ranges rg, int N, int M, SplitInfo sp) {
    static_assert(RT * NW * 16 == kMfmaRowsPerBlock, "256 rows per workgroup");
    constexpr int kMfmaRT = RT, kMfmaRowsPerWave = RT * 16, kBlock = NW * 64;      // (shadow the 4 x 4 constants of glhip_softmin_mfma.h)
    constexpr int NQ = WsumShape<MODE, D>::kNQ;
    constexpr int NA = WsumShape<MODE, D>::kNA;
    __shared__ uint4 tileX[(kTileX / 16) * 64];          // bf16 x 3 B operands, as in softmin_fwd_xdl_kernel
    __shared__ f32x4 tileQ[NQ * (kTileX / 64) * 16];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    int bx, b, split;
    workgroup_coords(sp, bx, b, split);
    const int ns = sp.n_splits;
    const int lk = lane >> 4;
    const int lj = lane & 15;

    int row_begin, row_end, q_begin, q_end;
    block_extent<SPARSE>(rg, N, kMfmaRowsPerBlock, row_begin, row_end, q_begin, q_end, bx);

    for (int row0 = row_begin; row0 < row_end; row0 += kMfmaRowsPerBlock) {
        float centre[D];
        // (A centre nearer the middle of the cloud — the mean of 8 rows of the block — halves the offsets of an unsorted cloud and was
        // tried for the gaussian modes in round 3 to widen the margin of the 1e6-point MMD gradient (9.7e-5 of a budget of 1e-4):
        // the same-law LOSS of that test went from 2.1e-5 to 2.6e-4 of the float64 value — the three terms of the norm are a
        // difference of 1e-3 of their size and owe their agreement to a common rounding pattern, which a per-block mean disturbs.
        // The first row it stays.)
        load_point<D, T>(prm.x, (long)b * N + row0, centre);
        // block-sparse: the wavefront -> rows map is rotated by the chunk index, so that the short or empty last wavefronts of the
        // partial chunks of successive row blocks fall on different SIMDs (as in glhip_softmin_x32.h)
        const int wslot = SPARSE ? ((wave + bx) & (NW - 1)) : wave;
        const int wave_row0 = row0 + wslot * kMfmaRowsPerWave;
        const bool wave_active = wave_row0 < row_end;

        uint4 A[kMfmaRT];
        f32x4 Cop[kMfmaRT];                // per-row constant added to every exponent
        f32x4 acc[kMfmaRT][NA];
#pragma unroll
        for (int rt = 0; rt < kMfmaRT; ++rt) {
            const int i = min(wave_row0 + rt * 16 + lj, row_end - 1);
            uint4 a = uint4{0u, 0u, 0u, 0u};
            if (lk < D) a = pack_a((to_f32<T>(prm.x[((long)b * N + i) * D + (lk < D ? lk : 0)]) - centre[lk < D ? lk : 0]) * prm.s2);
            else if (lk == 3) a = uint4{0x3F803F80u, 0x00003F80u, 0u, 0u};
            A[rt] = a;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ir = min(wave_row0 + rt * 16 + lk * 4 + r, row_end - 1);
                float xi[D];
                load_point<D, T>(prm.x, (long)b * N + ir, xi);
                float n2 = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float xt = xi[d] - centre[d];
                    n2 = __builtin_fmaf(xt, xt, n2);
                }
                const float ri = -0.5f * prm.s2 * n2;
                // -(LSE2 - r_i); in value-and-gradient mode `fwd` is a guess and tscale the margin that makes it an upper bound
                if (MODE == WS_SOFTMIN_BWD) Cop[rt][r] = ri - (prm.fwd[(long)b * N + ir] / prm.out_scale + prm.tscale);
                else Cop[rt][r] = ri;
            }
#pragma unroll
            for (int c = 0; c < NA; ++c) acc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        for (int q = q_begin + (SPARSE ? split : 0); q < q_end; q += (SPARSE ? ns : 1)) {
            int js, je;
            column_interval<SPARSE>(rg, M, q, split, ns, js, je);
            for (int j0 = js; j0 < je; j0 += kTileX) {
                const int n = min(kTileX, je - j0);
                const int npad = (n + 63) & ~63;
                __syncthreads();
                for (int t = tid; t < npad; t += kBlock) {
                    float rec[4] = {0.f, 0.f, 0.f, kNegBig};
                    float qv[NQ];
#pragma unroll
                    for (int c = 0; c < NQ; ++c) qv[c] = 0.f;
                    if (t < n) {
                        float yj[D];
                        load_point<D, T>(prm.y, (long)b * M + j0 + t, yj);
                        float n2 = 0.f;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            rec[d] = yj[d] - centre[d];
                            n2 = __builtin_fmaf(rec[d], rec[d], n2);
                        }
                        const float sj = prm.s[(long)b * M + j0 + t];
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
                    const int G = t >> 6, g = (t >> 4) & 3, j = t & 15;
                    uint4* base = &tileX[(t >> 4) * 64 + j];
#pragma unroll
                    for (int d = 0; d < 3; ++d) base[d * 16] = (d < D) ? pack_y(rec[d]) : uint4{0u, 0u, 0u, 0u};
                    base[48] = pack_h(rec[3]);
#pragma unroll
                    for (int c = 0; c < NQ; ++c)
                        reinterpret_cast<float*>(&tileQ[(c * (kTileX / 64) + G) * 16 + j])[g] = qv[c];
                }
                __syncthreads();
                if (!wave_active) continue;

                for (int G = 0; G < npad / 64; ++G) {
                    uint4 Bq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) Bq[g] = tileX[(G * 4 + g) * 64 + lane];
                    f32x4 Q[NQ];
#pragma unroll
                    for (int c = 0; c < NQ; ++c) Q[c] = tileQ[(c * (kTileX / 64) + G) * 16 + lj];
#pragma unroll
                    for (int rt = 0; rt < kMfmaRT; ++rt) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 w = exp2v(mfma_x(A[rt], Bq[g], Cop[rt]));
                            if (MODE == WS_SOFTMIN_BWD) {
#pragma unroll
                                for (int d = 0; d < D; ++d) fma4(acc[rt][d], w, Q[d][g]);
                                acc[rt][D] += w;
                            } else {
#pragma unroll
                                for (int c = 0; c < NQ; ++c) fma4(acc[rt][c], w, Q[c][g]);
                            }
                        }
                    }
                }
            }
        }

        if (wave_active) {
#pragma unroll
            for (int rt = 0; rt < kMfmaRT; ++rt) {
                // sum over the 16 column-lanes of each row
#pragma unroll
                for (int c = 0; c < NA; ++c) {
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[rt][c][r] += __shfl_xor(acc[rt][c][r], off, 64);
                    }
                }
                // the first lane of each 16-lane group writes the group's 4 rows (no per-lane select chains)
                if (lj == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = wave_row0 + rt * 16 + lk * 4 + r;
                        if (i < row_end) {
                            float a_[NA];
#pragma unroll
                            for (int c = 0; c < NA; ++c) a_[c] = acc[rt][c][r];
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
                                    const float gi = prm.g ? prm.g[(long)b * N + i] : 1.f;
                                    const float inv = (a_[D] > 0.f) ? 1.0f / a_[D] : 0.f;
#pragma unroll
                                    for (int d = 0; d < D; ++d) prm.gx[((long)b * N + i) * D + d] = gi * (xt[d] - a_[d] * inv);
                                    if (prm.out)   // value-and-gradient mode: the mass turns the guess into the exact soft-min
                                        prm.out[(long)b * N + i] = prm.fwd[(long)b * N + i] + prm.out_scale * (prm.tscale + fast_log2(a_[D]));
                                } else {
#pragma unroll
                                    for (int c = 0; c < NA; ++c) part[c] = a_[c];
                                }
                            } else if (MODE == WS_GAUSS_FWD) {
                                if (ns == 1) prm.out[(long)b * N + i] = a_[0];
                                else part[0] = a_[0];
                            } else if (MODE == WS_GAUSS_FWDGRAD) {
                                if (ns == 1) {
                                    prm.out[(long)b * N + i] = a_[D];
#pragma unroll
                                    for (int d = 0; d < D; ++d) prm.gx[((long)b * N + i) * D + d] = prm.gscale * (xt[d] * a_[D] - a_[d]);
                                } else {
#pragma unroll
                                    for (int d = 0; d < D; ++d) part[d] = prm.tscale * (xt[d] * a_[D] - a_[d]);
                                    part[D] = a_[D];
                                }
                            } else {
                                // sum_j v k (x - y) = xt S0 - S1
                                if (ns == 1) {
                                    const float gi = prm.g[(long)b * N + i] * prm.gscale;
#pragma unroll
                                    for (int d = 0; d < D; ++d) prm.gx[((long)b * N + i) * D + d] = gi * (xt[d] * a_[D] - a_[d]);
                                } else {
                                    // partials in the units of ConvOp<GAUSSIAN,...,BWD>::merge_row (scaled differences)
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
}

}  // namespace glhip
