// glhip_dist_xd.h — the DISTANCE reductions (soft-min with p = 1, laplacian and energy kernel products) for clouds of dimension
// 4 <= D <= 16, dense launches, with the squared distance formed on the matrix cores (round 5).
//
// The reference evaluates them with one KeOps formula whatever D is (`"Norm2(X-Y)"`, _legacy/sinkhorn_samples.py:316-319;
// `kernel_samples.py:71-82`).  Until round 4 everything beyond D = 3 went to the one-thread-per-row VALU kernel of glhip_generic.h:
// 2 D + 6 instructions per pair, 0.8e12 pairs/s at D = 4 ... 8 and 0.4e12 at D = 16 (tools/scratch measurement of round 5), a fifth
// of the D <= 3 kernels.  Here the scaled squared distance of a 32 x 32 block of pairs is the chain of MFMAs of glhip_softmin_xd.h,
//     d2_ij = |xs_i|^2 + |ys_j|^2 - 2 xs_i . ys_j,      xs = t (x - c), ys = t (y - c),
// with the K layout of that header (bf16 x 3: a squared distance is a difference of large terms and needs all 24 bits):
//     scalar item : y side [N1,N2,N3,1,1,1] (N = |ys_j|^2)       x side [1,1,1,n1,n2,n3] (n = |xs_i|^2)
//     coordinate d: y side pieces of ys_jd                        x side pieces of -2 xs_id
// and the VALU is left with, per pair: the clamp of utils.py:61 (v_med3), v_sqrt_f32 and
//     soft-min p = 1 : 2 sub, v_exp_f32, add (+ a 16-way max per block: exact online maximum)      laplacian : v_exp_f32, fma      energy : fma
// The per-column scalar (dual value / weight) is read from LDS as broadcast float4, as in xd_weighted_sum.
//
// ACCURACY.  d2 carries the absolute error of the expanded form, ~2^-23 t^2 R^2 (R = offset of the two points from the centre c of
// the launch): fine for d >~ R / 16, useless for near or coincident pairs — and the debiasing terms of a Sinkhorn loss are x against
// x.  Pairs with d2 < guard |xs_i|^2 (guard = 2^-8 by default: d < R_i / 16, as in glhip_dist_x32.h) are therefore re-evaluated on
// explicit differences of the points themselves (re-read from global memory: one block in ~60 at D = 4, none to speak of beyond
// D = 6, plus the diagonal of a self-term).  GLHIP_DIST_GUARD scales the threshold (0: off).
#pragma once

#include "glhip_dist_x32.h"
#include "glhip_softmin_xd.h"

namespace glhip {

template <int D>
struct DistXdShape {
    using S = XdShape<D, XL_BF16X3>;
    static constexpr int NM = S::NM, NBP = S::NBP;
    static constexpr int kTile = NBP <= 6 ? 256 : 128;      // columns per LDS tile: <= 28 KiB of records + 1 KiB of scalars
};

// exact squared distance (scaled by t^2, floored at clamp2) of row i and column j, from the points themselves
template <int D, typename T>
__device__ __forceinline__ float exact_d2(const T* __restrict__ x, const T* __restrict__ y, long i, long j, float t, float clamp2) {
    float xi[D], yj[D], e = 0.f;
    load_point<D, T>(x, i, xi);
    load_point<D, T>(y, j, yj);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float df = (xi[d] - yj[d]) * t;
        e = __builtin_fmaf(df, df, e);
    }
    return fmaxf(e, clamp2);
}

template <int MODE, int D, typename T, int NW>
__global__ void __launch_bounds__(NW * 64, 4)
dist_xd_kernel(DistParams<T> prm, int N, int M, SplitInfo sp) {
    using S = DistXdShape<D>;
    constexpr int NM = S::NM, NBP = S::NBP, kTileD = S::kTile;
    constexpr int kRowsPerBlock = NW * 32;
    constexpr int kThreads = NW * 64;
    __shared__ uint4 tile[kTileD * NBP];                              // [column group of 32][K block][column]
    __shared__ __attribute__((aligned(16))) float tileS[kTileD];      // S_j = log2(e) h_j (soft-min) | v_j (products)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, split = blockIdx.z, ns = sp.n_splits;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int rec0 = half * 32 + l31;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const T* xb = prm.x + (long)b * N * D;
    const T* yb = prm.y + (long)b * M * D;
    float centre[D];
    launch_centre<D, T>(prm.x, b, N, centre);

    const int row0 = blockIdx.x * kRowsPerBlock;
    const int wave_row0 = row0 + wave * 32;
    const bool wave_active = wave_row0 < N;
    const int i_lane = min(wave_row0 + l31, N - 1);
    uint4 X[NM];
    float nx;      // |xs_i|^2
    {
        float xi[D];
        load_point<D, T>(xb, i_lane, xi);
        float a[D];
        nx = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float xs = (xi[d] - centre[d]) * prm.t;
            nx = __builtin_fmaf(xs, xs, nx);
            a[d] = -2.f * xs;
        }
#pragma unroll
        for (int mm = 0; mm < NM; ++mm)
            X[mm] = select_u4(half != 0, xd_record_of<D, true>(2 * mm + 1, nx, a), xd_record_of<D, true>(2 * mm, nx, a));
    }
    const float thr = fmaxf(prm.guard * nx, 4.f * prm.clamp2);
    float m = kMinusHuge, ssum = 0.f;      // soft-min: exact running max and sum;  products: ssum only

    int js, je;
    {
        const int len = (((M + ns - 1) / ns) + 31) & ~31;
        js = min(M, split * len);
        je = min(M, js + len);
    }
    for (int j0 = js; j0 < je; j0 += kTileD) {
        const int n = min(kTileD, je - j0);
        const int npad = (n + 31) & ~31;
        __syncthreads();
        for (int t = tid; t < npad; t += kThreads) {
            float ys[D], n2 = 0.f, sj = (MODE == DM_SOFTMIN_P1) ? kNegBig : 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) ys[d] = 0.f;
            if (t < n) {
                float yj[D];
                load_point<D, T>(yb, j0 + t, yj);
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    ys[d] = (yj[d] - centre[d]) * prm.t;
                    n2 = __builtin_fmaf(ys[d], ys[d], n2);
                }
                const long col = (long)b * M + j0 + t;
                sj = prm.s[col];
                if (MODE == DM_SOFTMIN_P1) {
                    if (prm.pot) sj = __builtin_fmaf(prm.pot[col], prm.pot_scale, sj);
                    sj *= kLog2e;
                }
            }
            uint4* base = &tile[(t >> 5) * (32 * NBP) + (t & 31)];
#pragma unroll
            for (int r = 0; r < NBP; ++r) base[r * 32] = xd_record_of<D, false>(r, n2, ys);
            tileS[t] = sj;
        }
        __syncthreads();
        if (!wave_active) continue;

        for (int G = 0; G < npad / 32; ++G) {
            f32x16 d2 = xd_block<NM, NBP>(&tile[G * (32 * NBP)], rec0, X, zero16);
            const float* sg = &tileS[G * 32 + half * 4];
            float sv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 s4 = *reinterpret_cast<const float4*>(sg + q * 8);
                sv[q * 4] = s4.x; sv[q * 4 + 1] = s4.y; sv[q * 4 + 2] = s4.z; sv[q * 4 + 3] = s4.w;
            }
            // near pairs (d < R_i / 16: the partner then sits at the same offset from the centre) and everything below 4 x the floor of
            // utils.py:61, which only an exact value may decide
            if (prm.guard > 0.f && __any(min16(d2) < thr)) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int col = j0 + G * 32 + (k >> 2) * 8 + half * 4 + (k & 3);
                    if (d2[k] < thr && col < je) d2[k] = exact_d2<D, T>(xb, yb, i_lane, col, prm.t, prm.clamp2);
                }
            }
            if (MODE == DM_SOFTMIN_P1) {
                float u[16], um = kMinusHuge;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    u[k] = sv[k] - fast_sqrt(__builtin_amdgcn_fmed3f(d2[k], prm.clamp2, 3.0e38f));
                    um = fmaxf(um, u[k]);
                }
                um = fmaxf(um, __shfl_xor(um, 32, 64));
                const float mnew = fmaxf(m, um);
                float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 16; ++k) s4[k & 3] += fast_exp2(u[k] - mnew);
                ssum = __builtin_fmaf(ssum, fast_exp2(m - mnew), (s4[0] + s4[1]) + (s4[2] + s4[3]));
                m = mnew;
            } else {
                float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float dist = fast_sqrt(__builtin_amdgcn_fmed3f(d2[k], prm.clamp2, 3.0e38f));
                    if (MODE == DM_LAPLACIAN) a4[k & 3] = __builtin_fmaf(fast_exp2(-dist), sv[k], a4[k & 3]);
                    else a4[k & 3] = __builtin_fmaf(-dist, sv[k], a4[k & 3]);
                }
                ssum += (a4[0] + a4[1]) + (a4[2] + a4[3]);
            }
        }
    }

    if (wave_active) {
        const float s = ssum + __shfl_xor(ssum, 32, 64);     // the two halves hold the two 16-column halves of every block (same max)
        const int i = wave_row0 + l31;
        if (half == 0 && i < N) {
            const long idx = (long)b * N + i;
            if (MODE == DM_SOFTMIN_P1) {
                if (ns == 1) {
                    float f = prm.alpha * (prm.out_scale * (m + fast_log2(s)));
                    if (prm.prev) f = __builtin_fmaf(prm.beta, prm.prev[idx], f);
                    prm.out[idx] = f;
                } else {
                    float* dst = sp.workspace + split * sp.split_stride + idx * 2;
                    dst[0] = m;
                    dst[1] = s;
                }
            } else {
                if (ns == 1) prm.out[idx] = s;
                else sp.workspace[split * sp.split_stride + idx] = s;
            }
        }
    }
}

}  // namespace glhip
