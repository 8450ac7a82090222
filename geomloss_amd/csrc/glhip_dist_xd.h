// glhip_dist_xd.h — the DISTANCE reductions (soft-min with p = 1, laplacian and energy kernel products) for clouds of dimension
// 4 <= D <= 16, dense launches, with the squared distance formed on the matrix cores (round 5).
//
// The reference evaluates them with one KeOps formula whatever D is (`"Norm2(X-Y)"`, _legacy/sinkhorn_samples.py:316-319;
// `kernel_samples.py:71-82`).  Until round 4 everything beyond D = 3 went to the one-thread-per-row VALU kernel of glhip_generic.h:
// 2 D + 6 instructions per pair, 0.8e12 pairs/s at D = 4 ... 8 and 0.4e12 at D = 16 (measured in round 5), a fifth
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

// The work of one workgroup: row block bx of batch item b, column split `split`.
//   tile  [kTile * NBP] records: [column group of 32][K block][column];   tileS [kTile]: S_j = log2(e) h_j (soft-min) | v_j (products)
template <int MODE, int D, typename T, int NW>
__device__ __forceinline__ void dist_xd_body(const DistParams<T>& prm, int N, int M, const SplitInfo& sp, int bx, int b, int split,
                                             uint4* tile, float* tileS) {
    using S = DistXdShape<D>;
    constexpr int NM = S::NM, NBP = S::NBP, kTileD = S::kTile;
    constexpr int kRowsPerBlock = NW * 32;
    constexpr int kThreads = NW * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ns = sp.n_splits;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int rec0 = half * 32 + l31;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const T* xb = prm.x + (long)b * N * D;
    const T* yb = prm.y + (long)b * M * D;
    float centre[D];
    launch_centre<D, T>(prm.x, b, N, centre);

    const int row0 = bx * kRowsPerBlock;
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

template <int MODE, int D, typename T, int NW>
__global__ void __launch_bounds__(NW * 64, 4)
dist_xd_kernel(DistParams<T> prm, int N, int M, SplitInfo sp) {
    using S = DistXdShape<D>;
    __shared__ uint4 tile[S::kTile * S::NBP];
    __shared__ __attribute__((aligned(16))) float tileS[S::kTile];
    dist_xd_body<MODE, D, T, NW>(prm, N, M, sp, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, tile, tileS);
}

// Up to four independent dense p = 1 half-steps in ONE launch: the four updates of a Sinkhorn iteration (glhip_sinkhorn_iter4 with
// p = 1, round 5; any D <= 16).  grid = (max row blocks, B, n_splits * count); problem k = blockIdx.z / n_splits.
template <typename T>
struct DistMulti {
    DistParams<T> p[4];
    int N[4], M[4];
    long ws_stride;     // floats of split workspace per problem
    int count;
};

template <int D, typename T, int NW>
__global__ void __launch_bounds__(NW * 64, 4)
dist_xd_multi_kernel(DistMulti<T> m, SplitInfo sp) {
    using S = DistXdShape<D>;
    __shared__ uint4 tile[S::kTile * S::NBP];
    __shared__ __attribute__((aligned(16))) float tileS[S::kTile];
    const int k = blockIdx.z / sp.n_splits;
    const int split = blockIdx.z - k * sp.n_splits;
    const int N = m.N[k], M = m.M[k];
    if ((int)blockIdx.x * (NW * 32) >= N) return;
    SplitInfo spk = sp;
    spk.workspace += k * m.ws_stride;
    spk.split_stride = (long)gridDim.y * N * 2;   // this problem's own row count
    dist_xd_body<DM_SOFTMIN_P1, D, T, NW>(m.p[k], N, M, spk, (int)blockIdx.x, (int)blockIdx.y, split, tile, tileS);
}


// ---- gradients with respect to the row points (round 5) ----------------------------------------------------------------------------
//   energy    : d/dx_i sum_j -|x_i - y_j| v_j                 = -      sum_j v_j (x_i - y_j) / |x_i - y_j|
//   laplacian : d/dx_i sum_j exp(-|x_i - y_j| / blur) v_j     = -1/blur sum_j v_j k_ij (x_i - y_j) / |x_i - y_j|
//   soft-min  : d f_i / d x_i (p = 1)                         =         sum_j P_ij (x_i - y_j) / |x_i - y_j| / sum_j P_ij,   P_ij = exp(h_j - |x_i - y_j| / eps - LSE_i)
// all of the form  xs_i S0 - S1  with  S0 = sum_j w_ij,  S1 = sum_j w_ij ys_j  in the scaled, centred coordinates of the forward kernel
// (the direction (xs - ys) / |xs - ys| does not depend on the scale), and w_ij = 0 inside the clamp of utils.py:61, like autograd
// through `sqrt(clamp_min(d2, 1e-8))`.  The squared distances come from the same MFMA chain as the forward kernel (exact near pairs
// included); the weights cost one v_rsq_f32 (+ one v_exp_f32) per pair and the sums D + 1 v_fma_f32 with the column coordinates read
// back from LDS as broadcast float4 — the one-thread-per-row kernel these replace spends 2 D + 8 instructions per pair on explicit
// differences.  Partials of a column split: (xs S0 - S1)_d [, sum_j P_ij]: the formats of ConvOp<.., 1> / SoftminBwdOp<.., 1, ..>.
template <typename T>
struct DistXdGradParams {
    DistParams<T> d;      // x, y, s (h | v), t, clamp2, out_scale (soft-min: -eps ln 2), guard; out / pot / prev unused
    const float* fwd;     // soft-min: the forward values f_i (B,N)
    const float* g;       // (B,N) incoming gradient
    float* gx;            // (B,N,D)
    float gscale;         // products: -1 (energy), -1 / blur (laplacian)
};

template <int MODE, int D, typename T, int NW>
__global__ void __launch_bounds__(NW * 64, 4)
dist_xd_grad_kernel(DistXdGradParams<T> gp, int N, int M, SplitInfo sp) {
    using S = DistXdShape<D>;
    constexpr int NM = S::NM, NBP = S::NBP, kTileD = 128;
    constexpr int kThreads = NW * 64;
    constexpr bool SM = MODE == DM_SOFTMIN_P1;
    const DistParams<T>& prm = gp.d;
    __shared__ uint4 tile[kTileD * NBP];
    __shared__ __attribute__((aligned(16))) float tileS[kTileD];
    __shared__ __attribute__((aligned(16))) float tileY[D * kTileD];      // [coordinate][column]: ys_j, read back as broadcast float4

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

    const int wave_row0 = blockIdx.x * (NW * 32) + wave * 32;
    const bool wave_active = wave_row0 < N;
    const int i_lane = min(wave_row0 + l31, N - 1);
    uint4 X[NM];
    float nx, lse2 = 0.f;
    {
        float xi[D], a[D];
        load_point<D, T>(xb, i_lane, xi);
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
        if (SM) lse2 = gp.fwd[(long)b * N + i_lane] / prm.out_scale;      // base-2 log-sum-exp of the row
    }
    const float thr = fmaxf(prm.guard * nx, 4.f * prm.clamp2);
    float S0 = 0.f, mass = 0.f, S1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) S1[d] = 0.f;

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
            float ys[D], n2 = 0.f, sj = SM ? kNegBig : 0.f;
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
                sj = prm.s[(long)b * M + j0 + t];
                if (SM) sj *= kLog2e;
            }
            uint4* base = &tile[(t >> 5) * (32 * NBP) + (t & 31)];
#pragma unroll
            for (int r = 0; r < NBP; ++r) base[r * 32] = xd_record_of<D, false>(r, n2, ys);
            tileS[t] = sj;
#pragma unroll
            for (int d = 0; d < D; ++d) tileY[d * kTileD + t] = ys[d];
        }
        __syncthreads();
        if (!wave_active) continue;

        for (int G = 0; G < npad / 32; ++G) {
            f32x16 d2 = xd_block<NM, NBP>(&tile[G * (32 * NBP)], rec0, X, zero16);
            // Near pairs.  Not only their distance needs explicit differences: so does their DIRECTION — (xs_i - ys_j) of a pair 1e-4 apart
            // is lost in  xs_i S0 - S1  (terms of size |xs| w, w = 1 / d huge).  Such a pair is taken out of the sums below (its squared
            // distance becomes "infinite": weight 0) and its whole contribution w (xs_i - ys_j) goes into -S1 from the points themselves.
            if (prm.guard > 0.f && __any(min16(d2) < thr)) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int cl = G * 32 + (k >> 2) * 8 + half * 4 + (k & 3);
                    if (d2[k] < thr && j0 + cl < je) {
                        float xi[D], yj[D], df[D], e = 0.f;
                        load_point<D, T>(xb, i_lane, xi);
                        load_point<D, T>(yb, j0 + cl, yj);
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            df[d] = (xi[d] - yj[d]) * prm.t;
                            e = __builtin_fmaf(df[d], df[d], e);
                        }
                        const bool far = e > prm.clamp2;
                        const float m = fmaxf(e, prm.clamp2), rs = fast_rsq(m), sj = tileS[cl];
                        float wk;
                        if (MODE == DM_ENERGY) {
                            wk = sj * rs;
                        } else if (MODE == DM_LAPLACIAN) {
                            wk = sj * fast_exp2(-m * rs) * rs;
                        } else {
                            const float P = fast_exp2(sj - m * rs - lse2);
                            mass += P;
                            wk = P * rs;
                        }
                        if (far) {
#pragma unroll
                            for (int d = 0; d < D; ++d) S1[d] = __builtin_fmaf(-wk, df[d], S1[d]);
                        }
                        d2[k] = 3.0e38f;
                    }
                }
            }
            float w[16];
            const float* sg = &tileS[G * 32 + half * 4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 s4 = *reinterpret_cast<const float4*>(sg + q * 8);
                const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dd = d2[q * 4 + r];
                    const bool far = dd > prm.clamp2;                    // inside the clamp the distance is constant: no direction
                    const float m = __builtin_amdgcn_fmed3f(dd, prm.clamp2, 3.0e38f);
                    const float rs = fast_rsq(m);
                    float wk;
                    if (MODE == DM_ENERGY) {
                        wk = sv[r] * rs;
                    } else if (MODE == DM_LAPLACIAN) {
                        wk = sv[r] * fast_exp2(-m * rs) * rs;
                    } else {
                        const float P = fast_exp2(sv[r] - m * rs - lse2);
                        mass += P;
                        wk = P * rs;
                    }
                    w[q * 4 + r] = far ? wk : 0.f;
                }
            }
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) s4[k & 3] += w[k];
            S0 += (s4[0] + s4[1]) + (s4[2] + s4[3]);
            const float* yg = &tileY[G * 32 + half * 4];
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 y4 = *reinterpret_cast<const float4*>(yg + d * kTileD + q * 8);
                    S1[d] = __builtin_fmaf(w[q * 4], y4.x, S1[d]);
                    S1[d] = __builtin_fmaf(w[q * 4 + 1], y4.y, S1[d]);
                    S1[d] = __builtin_fmaf(w[q * 4 + 2], y4.z, S1[d]);
                    S1[d] = __builtin_fmaf(w[q * 4 + 3], y4.w, S1[d]);
                }
            }
        }
    }

    if (wave_active) {
        S0 += __shfl_xor(S0, 32, 64);        // the two halves hold the two 16-column halves of every block
        mass += __shfl_xor(mass, 32, 64);
#pragma unroll
        for (int d = 0; d < D; ++d) S1[d] += __shfl_xor(S1[d], 32, 64);
        const int i = wave_row0 + l31;
        if (half == 0 && i < N) {
            const long idx = (long)b * N + i;
            float xi[D];
            load_point<D, T>(xb, i, xi);
            float dir[D];
#pragma unroll
            for (int d = 0; d < D; ++d) dir[d] = __builtin_fmaf((xi[d] - centre[d]) * prm.t, S0, -S1[d]);      // (xs S0 - S1)_d
            if (ns == 1) {
                const float gi = gp.g ? gp.g[idx] : 1.f;
                const float f = SM ? (mass > 0.f ? gi / mass : 0.f) : gi * gp.gscale;
#pragma unroll
                for (int d = 0; d < D; ++d) gp.gx[idx * D + d] = f * dir[d];
            } else {
                float* part = sp.workspace + split * sp.split_stride + idx * (SM ? D + 1 : D);
#pragma unroll
                for (int d = 0; d < D; ++d) part[d] = dir[d];
                if (SM) part[D] = mass;
            }
        }
    }
}

}  // namespace glhip
