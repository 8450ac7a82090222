// glhip_generic.h — any-dimension fallback (D >= 4) for the same reductions, and the
// row-wise soft-min of an explicit dense cost matrix.
//
// Generic-D kernel: one thread per row, columns in tiles of kJT, feature axis in chunks of kDC.
// The (kJT x kDC) slab of y is staged in LDS and read back with broadcast ds_read_b128; each thread
// keeps kJT partial squared distances in VGPRs.  Cost is ~2 VALU ops per (pair, dimension): this is
// the slow path, the library is tuned for D <= 3.
#pragma once

#include "glhip_common.h"

namespace glhip {

constexpr int kJT = 16;   // columns per generic tile
constexpr int kDC = 16;   // feature dimensions per chunk
constexpr int kGenericMaxGradD = 64;

enum GenericMode { GM_SOFTMIN_P2 = 0, GM_SOFTMIN_P1 = 1, GM_GAUSS = 2, GM_LAPLACE = 3, GM_ENERGY = 4 };

template <typename T>
struct GenericParams {
    const T* x;         // (B,N,D)
    const T* y;         // (B,M,D)
    const float* s;     // (B,M): h (softmin) or v (kernel product)
    float* out;         // fwd (B,N)
    const float* fwd;   // softmin bwd: saved forward (B,N)
    const float* g;     // bwd (B,N)
    float* gx;          // bwd (B,N,D)
    float dscale;       // factor on d2 (p=2 / gaussian) or on sqrt(d2) (p=1 / laplacian), base-2 units
    float out_scale;    // softmin: -eps ln2
    float gscale;       // kernel bwd: factor on the accumulated direction sum
    float clamp2;       // floor on squared distances under a square root (1e-8, utils.py:61)
};

template <int MODE>
__device__ __forceinline__ float generic_value(float d2, float sj, float dscale, float clamp2) {
    // softmin: exponent u_ij (base 2);  kernels: k_ij (unweighted)
    if (MODE == GM_SOFTMIN_P2) return __builtin_fmaf(-d2, dscale, sj);
    if (MODE == GM_SOFTMIN_P1) return __builtin_fmaf(-fast_sqrt(fmaxf(d2, clamp2)), dscale, sj);
    if (MODE == GM_GAUSS) return fast_exp2(-d2 * dscale);
    if (MODE == GM_LAPLACE) return fast_exp2(-fast_sqrt(fmaxf(d2, clamp2)) * dscale);
    return -fast_sqrt(fmaxf(d2, clamp2));
}

template <int MODE, bool BWD, bool SPARSE, typename T>
__global__ void __launch_bounds__(kBlock)
generic_kernel(GenericParams<T> p, Ranges rg, int N, int M, int D) {
    constexpr bool SOFTMIN = (MODE == GM_SOFTMIN_P2 || MODE == GM_SOFTMIN_P1);
    __shared__ __attribute__((aligned(16))) float ytile[kJT][kDC];
    __shared__ float stile[kJT];

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
        row_begin = blockIdx.x * kBlock;
        row_end = min(N, row_begin + kBlock);
        q_begin = 0;
        q_end = 1;
    }
    const int nchunks = (D + kDC - 1) / kDC;

    for (int row0 = row_begin; row0 < row_end; row0 += kBlock) {
        const int i = min(row0 + tid, row_end - 1);
        const T* xi = p.x + ((long)b * N + i) * D;
        float m = kNegBig, ssum = 0.f, accv = 0.f;
        float lse2 = 0.f;
        float gacc[BWD ? kGenericMaxGradD : 1];
        if (BWD) {
#pragma unroll
            for (int d = 0; d < kGenericMaxGradD; ++d) gacc[d] = 0.f;
            if (SOFTMIN) lse2 = p.fwd[(long)b * N + i] / p.out_scale;
        }

        for (int q = q_begin; q < q_end; ++q) {
            const int js = SPARSE ? rg.redranges_j[2 * q] : 0;
            const int je = SPARSE ? rg.redranges_j[2 * q + 1] : M;
            for (int j0 = js; j0 < je; j0 += kJT) {
                float d2[kJT];
#pragma unroll
                for (int jj = 0; jj < kJT; ++jj) d2[jj] = 0.f;
                float sj[kJT];

                for (int ch = 0; ch < nchunks; ++ch) {
                    const int d0 = ch * kDC;
                    __syncthreads();
                    {
                        const int jj = tid / kDC, dd = tid % kDC;
                        const int j = j0 + jj, d = d0 + dd;
                        ytile[jj][dd] = (j < je && d < D) ? to_f32<T>(p.y[((long)b * M + j) * D + d]) : 0.f;
                        if (ch == 0 && tid < kJT) {
                            const int jt = j0 + tid;
                            float sv = SOFTMIN ? kNegBig : 0.f;
                            if (jt < je) sv = SOFTMIN ? p.s[(long)b * M + jt] * kLog2e : p.s[(long)b * M + jt];
                            stile[tid] = sv;
                        }
                    }
                    __syncthreads();
                    float xd[kDC];
#pragma unroll
                    for (int dd = 0; dd < kDC; ++dd) xd[dd] = (d0 + dd < D) ? to_f32<T>(xi[d0 + dd]) : 0.f;
#pragma unroll
                    for (int jj = 0; jj < kJT; ++jj) {
#pragma unroll
                        for (int dd = 0; dd < kDC; ++dd) {
                            const float df = xd[dd] - ytile[jj][dd];
                            d2[jj] = __builtin_fmaf(df, df, d2[jj]);
                        }
                    }
                    if (ch == 0) {
#pragma unroll
                        for (int jj = 0; jj < kJT; ++jj) sj[jj] = stile[jj];
                    }
                }

                if (!BWD) {
                    if (SOFTMIN) {
                        float u[kJT];
#pragma unroll
                        for (int jj = 0; jj < kJT; ++jj) u[jj] = generic_value<MODE>(d2[jj], sj[jj], p.dscale, p.clamp2);
                        float cm = u[0];
#pragma unroll
                        for (int jj = 1; jj < kJT; ++jj) cm = fmaxf(cm, u[jj]);
                        const float mn = fmaxf(m, cm);
                        ssum *= fast_exp2(m - mn);
                        m = mn;
#pragma unroll
                        for (int jj = 0; jj < kJT; ++jj) ssum += fast_exp2(u[jj] - mn);
                    } else {
#pragma unroll
                        for (int jj = 0; jj < kJT; ++jj)
                            accv = __builtin_fmaf(generic_value<MODE>(d2[jj], 0.f, p.dscale, p.clamp2), sj[jj], accv);
                    }
                } else {
                    // direction weights w_j, then a second sweep over the feature chunks
                    float w[kJT];
#pragma unroll
                    for (int jj = 0; jj < kJT; ++jj) {
                        const float rs = (d2[jj] > p.clamp2) ? fast_rsq(d2[jj]) : 0.f;
                        if (SOFTMIN) {
                            const float pij = fast_exp2(generic_value<MODE>(d2[jj], sj[jj], p.dscale, p.clamp2) - lse2);
                            ssum += pij;
                            w[jj] = (MODE == GM_SOFTMIN_P2) ? pij : pij * rs;
                        } else if (MODE == GM_GAUSS) {
                            w[jj] = sj[jj] * generic_value<MODE>(d2[jj], 0.f, p.dscale, p.clamp2);
                        } else if (MODE == GM_LAPLACE) {
                            w[jj] = sj[jj] * rs * generic_value<MODE>(d2[jj], 0.f, p.dscale, p.clamp2);
                        } else {
                            w[jj] = sj[jj] * rs;
                        }
                    }
#pragma unroll
                    for (int ch = 0; ch < kGenericMaxGradD / kDC; ++ch) {
                        const int d0 = ch * kDC;
                        if (d0 < D) {
                            if (nchunks > 1) {   // the slab in LDS is the last chunk: restage
                                __syncthreads();
                                const int jj = tid / kDC, dd = tid % kDC;
                                const int j = j0 + jj, d = d0 + dd;
                                ytile[jj][dd] = (j < je && d < D) ? to_f32<T>(p.y[((long)b * M + j) * D + d]) : 0.f;
                                __syncthreads();
                            }
#pragma unroll
                            for (int dd = 0; dd < kDC; ++dd) {
                                const float xv = (d0 + dd < D) ? to_f32<T>(xi[d0 + dd]) : 0.f;
                                float a = gacc[d0 + dd];
#pragma unroll
                                for (int jj = 0; jj < kJT; ++jj) a = __builtin_fmaf(w[jj], xv - ytile[jj][dd], a);
                                gacc[d0 + dd] = a;
                            }
                        }
                    }
                }
            }
        }

        const int irow = row0 + tid;
        if (irow < row_end) {
            if (!BWD) {
                p.out[(long)b * N + irow] = SOFTMIN ? p.out_scale * (m + fast_log2(ssum)) : accv;
            } else {
                float f = p.g[(long)b * N + irow];
                if (SOFTMIN) f *= (ssum > 0.f) ? 1.0f / ssum : 0.f;
                else f *= p.gscale;
#pragma unroll
                for (int d = 0; d < kGenericMaxGradD; ++d)
                    if (d < D) p.gx[((long)b * N + irow) * D + d] = f * gacc[d];
            }
        }
    }
}

// ---- explicit dense cost matrix: out_i = out_scale * log2 sum_j 2^(log2e h_j - s2 C_ij) ----------
// One wavefront reduces kDenseRows rows at once (they share the h_j loads); lanes stride over the
// columns with 16-byte loads, then merge their (max, sum) pairs with a 6-step xor butterfly.
// This kernel is what the "4 bytes per pair" dense-equivalent roofline literally describes.
constexpr int kDenseRows = 4;

__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    s = s * fast_exp2(m - mn) + s2 * fast_exp2(m2 - mn);
    m = mn;
}

template <bool VEC4>
__global__ void __launch_bounds__(kBlock)
softmin_dense_kernel(const float* __restrict__ C, const float* __restrict__ h, float* __restrict__ out,
                     int N, int M, float s2, float out_scale) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int row0 = wave * kDenseRows;
    if (row0 >= N) return;
    const float* hb = h + (long)b * M;
    const float* rows[kDenseRows];
#pragma unroll
    for (int r = 0; r < kDenseRows; ++r) rows[r] = C + ((long)b * N + min(row0 + r, N - 1)) * M;

    float m[kDenseRows], s[kDenseRows];
#pragma unroll
    for (int r = 0; r < kDenseRows; ++r) { m[r] = kNegBig; s[r] = 0.f; }

    if (VEC4) {
        for (int j = lane * 4; j < M; j += 64 * 4) {
            const float4 hv = *reinterpret_cast<const float4*>(hb + j);
            const float hh[4] = {hv.x * kLog2e, hv.y * kLog2e, hv.z * kLog2e, hv.w * kLog2e};
            float4 cv[kDenseRows];
#pragma unroll
            for (int r = 0; r < kDenseRows; ++r) cv[r] = *reinterpret_cast<const float4*>(rows[r] + j);
#pragma unroll
            for (int r = 0; r < kDenseRows; ++r) {
                const float u0 = __builtin_fmaf(-cv[r].x, s2, hh[0]);
                const float u1 = __builtin_fmaf(-cv[r].y, s2, hh[1]);
                const float u2 = __builtin_fmaf(-cv[r].z, s2, hh[2]);
                const float u3 = __builtin_fmaf(-cv[r].w, s2, hh[3]);
                const float mn = fmaxf(fmaxf(fmaxf(u0, u1), fmaxf(u2, u3)), m[r]);
                s[r] = s[r] * fast_exp2(m[r] - mn) + (fast_exp2(u0 - mn) + fast_exp2(u1 - mn)) +
                       (fast_exp2(u2 - mn) + fast_exp2(u3 - mn));
                m[r] = mn;
            }
        }
    } else {
        for (int j = lane; j < M; j += 64) {
            const float hh = hb[j] * kLog2e;
#pragma unroll
            for (int r = 0; r < kDenseRows; ++r) {
                const float u = __builtin_fmaf(-rows[r][j], s2, hh);
                const float mn = fmaxf(u, m[r]);
                s[r] = s[r] * fast_exp2(m[r] - mn) + fast_exp2(u - mn);
                m[r] = mn;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kDenseRows; ++r) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float m2 = __shfl_xor(m[r], off, 64);
            const float s2_ = __shfl_xor(s[r], off, 64);
            lse_merge(m[r], s[r], m2, s2_);
        }
        if (lane == 0 && row0 + r < N) out[(long)b * N + row0 + r] = out_scale * (m[r] + fast_log2(s[r]));
    }
}

}  // namespace glhip
