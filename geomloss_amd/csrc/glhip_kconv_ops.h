// glhip_kconv_ops.h — row operators for kernel-matrix x vector products
//   out_i = sum_j k(x_i, y_j) v_j,  k in {gaussian, laplacian, energy}
// and for their gradient with respect to the row points.  Plugged into mapreduce_kernel.
//
// All three kernels are radial, and all three are evaluated on explicit differences
// (xs_d - ys_d) of centred, pre-scaled coordinates — no |x|^2 - 2x.y + |y|^2 expansion, so the
// laplacian / energy kernels do not inherit the cancellation error of the dense reference code
// near x = y (utils.py:42-61 clamps that error at 1e-8 instead).
//   gaussian : t = sqrt(log2(e)/2) / blur,  k = 2^(-|xs-ys|^2)
//   laplacian: t = log2(e) / blur,          k = 2^(-|xs-ys|)
//   energy   : t = 1,                       k = -|xs-ys|
// with |.| floored at 1e-4 t, the clamp of the reference's dense `distances` (utils.py:61).
// LDS record: { ys_0 .. ys_{D-1}, v_j }.
#pragma once

#include "glhip_mapreduce.h"

namespace glhip {

template <typename T>
struct ConvParams {
    const T* x;        // (B,N,D)
    const T* y;        // (B,M,D)
    const float* v;    // (B,M)
    float* out;        // fwd: (B,N)
    const float* g;    // bwd: (B,N)
    float* gx;         // bwd: (B,N,D)
    float t;           // coordinate pre-scale
    float gscale;      // bwd: factor applied to the accumulated direction sum
    float clamp2;      // laplacian / energy: floor on the scaled squared distance, 1e-8 * t^2 (utils.py:61)
};

// FAMILY: |.| as m * rsq(m), the arithmetic of the product-and-gradient mode below (which needs rsq for the direction and
// cannot afford a second transcendental) instead of sqrt(m): about twice the rounding noise of sqrt, but the SAME rounding
// as that mode, which is what the three terms of one kernel norm need (kernel_samples._kernel_operators).
template <int KIND, bool FAMILY = false>
__device__ __forceinline__ float radial_kernel(float d2, float clamp2) {
    if (KIND == GLHIP_GAUSSIAN) return fast_exp2(-d2);
    const float m = fmaxf(d2, clamp2);
    const float dist = FAMILY ? m * fast_rsq(m) : fast_sqrt(m);
    if (KIND == GLHIP_LAPLACIAN) return fast_exp2(-dist);
    return -dist;
}

// MODE 0: product.  MODE 1: gradient, scaled by the incoming g_i.  MODE 2: product AND unit gradient
// in one pass (acc = { direction sum, product }; gx = gscale * direction sum, no g) — the caller
// scales the saved unit gradient in backward, so the gradient reduction never runs.  MODE 3: product,
// rounded exactly like the product of MODE 2 (GLHIP_FLAG_GRAD_FAMILY).
template <int KIND, int D_, int R, typename T, int MODE>
struct ConvOp {
    static constexpr int kDim = D_;
    static constexpr int kRows = R;
    static constexpr bool BWD = MODE == 1 || MODE == 2;
    static constexpr bool BOTH = MODE == 2;
    static constexpr bool FAMILY = MODE == 3;
    static constexpr int kAcc = BOTH ? D_ + 1 : (BWD ? D_ : 1);
    using Params = ConvParams<T>;
    struct RowState {
        float a[R][D_];
        float acc[R][kAcc];
        float clamp2;
    };

    static __device__ __forceinline__ void load_centre(const Params& p, int b, int N, int row0, float (&c)[D_]) {
        load_point<D_, T>(p.x, (long)b * N + row0, c);
    }

    static __device__ __forceinline__ void init_rows(const Params& p, int b, int N, int row0, int row_end,
                                                     int tid, const float (&c)[D_], RowState& st) {
        st.clamp2 = p.clamp2;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = min(row0 + r * kBlock + tid, row_end - 1);
            float xi[D_];
            load_point<D_, T>(p.x, (long)b * N + i, xi);
#pragma unroll
            for (int d = 0; d < D_; ++d) st.a[r][d] = (xi[d] - c[d]) * p.t;
#pragma unroll
            for (int d = 0; d < kAcc; ++d) st.acc[r][d] = 0.f;
        }
    }

    static __device__ __forceinline__ Rec<D_> make_record(const Params& p, int b, int M, int j, const float (&c)[D_]) {
        float yj[D_];
        load_point<D_, T>(p.y, (long)b * M + j, yj);
        Rec<D_> rec;
#pragma unroll
        for (int d = 0; d < D_; ++d) rec.c[d] = (yj[d] - c[d]) * p.t;
        rec_tail<D_>(rec) = p.v[(long)b * M + j];
        if (D_ == 2) rec.c[3] = 0.f;
        return rec;
    }

    static __device__ __forceinline__ Rec<D_> neutral_record() {
        Rec<D_> rec;
#pragma unroll
        for (int d = 0; d < D_; ++d) rec.c[d] = 0.f;
        rec_tail<D_>(rec) = 0.f;   // zero weight: contributes nothing
        if (D_ == 2) rec.c[3] = 0.f;
        return rec;
    }

    static __device__ __forceinline__ void consume(RowState& st, const Rec<D_>* __restrict__ recs) {
        Rec<D_> rc[kChunk];
#pragma unroll
        for (int c = 0; c < kChunk; ++c) rc[c] = recs[c];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int c = 0; c < kChunk; ++c) {
                float df[D_];
                float d2 = 0.f;
#pragma unroll
                for (int d = 0; d < D_; ++d) {
                    df[d] = st.a[r][d] - rc[c].c[d];
                    d2 = __builtin_fmaf(df[d], df[d], d2);
                }
                const float vj = rec_tail<D_>(rc[c]);
                if (!BWD) {
                    st.acc[r][0] = __builtin_fmaf(radial_kernel<KIND, FAMILY>(d2, st.clamp2), vj, st.acc[r][0]);
                } else {
                    // weight of the direction (xs - ys):  gaussian k ; laplacian k/|.| ; energy 1/|.|
                    float w;
                    if (KIND == GLHIP_GAUSSIAN) {
                        const float k = fast_exp2(-d2);
                        if (BOTH) st.acc[r][D_] = __builtin_fmaf(k, vj, st.acc[r][D_]);   // the product mode's fma
                        w = vj * k;
                    } else if (!BOTH) {
                        const float rs = (d2 > st.clamp2) ? fast_rsq(d2) : 0.f;
                        w = (KIND == GLHIP_LAPLACIAN) ? vj * rs * fast_exp2(-d2 * rs) : vj * rs;
                    } else {
                        // the product needs the clamped distance itself (utils.py:61), the direction
                        // still vanishes inside the clamp
                        const bool far = d2 > st.clamp2;
                        const float m = fmaxf(d2, st.clamp2);
                        const float rs = fast_rsq(m);
                        const float dist = m * rs;            // radial_kernel<KIND, true>, bit for bit
                        if (KIND == GLHIP_LAPLACIAN) {
                            const float k = fast_exp2(-dist);
                            st.acc[r][D_] = __builtin_fmaf(k, vj, st.acc[r][D_]);
                            w = far ? vj * k * rs : 0.f;
                        } else {
                            st.acc[r][D_] = __builtin_fmaf(-dist, vj, st.acc[r][D_]);
                            w = far ? vj * rs : 0.f;
                        }
                    }
#pragma unroll
                    for (int d = 0; d < D_; ++d) st.acc[r][d] = __builtin_fmaf(w, df[d], st.acc[r][d]);
                }
            }
        }
    }

    static __device__ __forceinline__ void finish_rows(const Params& p, int b, int N, int row0, int row_end,
                                                       int tid, const float (&)[D_], RowState& st) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = row0 + r * kBlock + tid;
            if (i < row_end) {
                if (!BWD) {
                    p.out[(long)b * N + i] = st.acc[r][0];
                } else {
                    if (BOTH) p.out[(long)b * N + i] = st.acc[r][D_];
                    const float gi = BOTH ? p.gscale : p.g[(long)b * N + i] * p.gscale;
#pragma unroll
                    for (int d = 0; d < D_; ++d) p.gx[((long)b * N + i) * D_ + d] = gi * st.acc[r][d];
                }
            }
        }
    }

    // column splits: plain partial sums
    static constexpr int kPartial = kAcc;
    static __device__ __forceinline__ void store_partial(const RowState& st, int r, float* dst) {
#pragma unroll
        for (int d = 0; d < kPartial; ++d) dst[d] = st.acc[r][d];
    }
    static __device__ __forceinline__ void merge_row(const Params& p, int b, int N, int i, const float (&)[D_],
                                                     const float* part, int ns, long stride) {
        float acc[kPartial];
#pragma unroll
        for (int d = 0; d < kPartial; ++d) acc[d] = 0.f;
        for (int k = 0; k < ns; ++k) {
#pragma unroll
            for (int d = 0; d < kPartial; ++d) acc[d] += part[k * stride + d];
        }
        if (!BWD) {
            p.out[(long)b * N + i] = acc[0];
        } else {
            if (BOTH) p.out[(long)b * N + i] = acc[D_];
            const float gi = BOTH ? p.gscale : p.g[(long)b * N + i] * p.gscale;
#pragma unroll
            for (int d = 0; d < D_; ++d) p.gx[((long)b * N + i) * D_ + d] = gi * acc[d];
        }
    }
};

// merge of the column splits of WS_GAUSS_FWDGRAD (glhip_wsum_mfma.h): partial = { t (xt S0 - S1)_d , S0 }
template <int D_, typename T>
struct GaussFwdGradMerge {
    static constexpr int kDim = D_;
    static constexpr int kRows = 1;
    static constexpr int kPartial = D_ + 1;
    using Params = ConvParams<T>;
    static __device__ __forceinline__ void load_centre(const Params&, int, int, int, float (&)[D_]) {}
    static __device__ __forceinline__ void merge_row(const Params& p, int b, int N, int i, const float (&)[D_],
                                                     const float* part, int ns, long stride) {
        float acc[kPartial];
#pragma unroll
        for (int d = 0; d < kPartial; ++d) acc[d] = 0.f;
        for (int k = 0; k < ns; ++k) {
#pragma unroll
            for (int d = 0; d < kPartial; ++d) acc[d] += part[k * stride + d];
        }
        p.out[(long)b * N + i] = acc[D_];
#pragma unroll
        for (int d = 0; d < D_; ++d) p.gx[((long)b * N + i) * D_ + d] = p.gscale * acc[d];
    }
};

}  // namespace glhip
