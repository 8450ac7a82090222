// glhip_softmin_ops.h — row operators for the soft-C-transform (log-sum-exp over columns)
// and its gradient with respect to the row points.  Plugged into mapreduce_kernel.
//
// Everything is evaluated in base 2 so that the inner loop is FMA chain -> v_exp_f32:
//   u_ij = log2(e) * ( h_j - C(x_i, y_j) / eps ),   LSE2_i = log2 sum_j 2^u_ij,
//   out_i = -eps * ln(2) * LSE2_i.
//
// Two evaluation forms of u_ij:
//   EXPANDED (p = 2 only, default):  with xt = x - c, yt = y - c, s2 = log2(e)/eps,
//       u_ij = r_i + [ H_j + a_i . yt_j ],   a_i = s2 * xt_i,  r_i = -s2/2 |xt_i|^2,
//       H_j = log2(e) h_j - s2/2 |yt_j|^2                       -> 3 FMAs per pair (D = 3)
//     r_i is constant along the row and is added after the reduction.
//   DIRECT (p = 1, or p = 2 with GLHIP_FLAG_DIRECT): coordinates pre-scaled by t,
//       p = 2: t = sqrt(s2/2), u_ij = H_j - sum_d (xs_d - ys_d)^2
//       p = 1: t = s2,         u_ij = H_j - sqrt( max( sum_d (xs_d - ys_d)^2, 1e-8 t^2 ) )
//              (the floor is the clamp of the reference's dense cost, utils.py:61)
//
// The running maximum is exact (updated every kChunk columns), so no threshold / rescale branch exists.
#pragma once

#include "glhip_mapreduce.h"

namespace glhip {

template <typename T>
struct SoftminParams {
    const T* x;        // (B,N,D)
    const T* y;        // (B,M,D)
    const float* h;    // (B,M)
    float* out;        // fwd: result (B,N);          bwd: unused
    const float* fwd;  // bwd: saved forward result (B,N)
    const float* g;    // bwd: grad_out (B,N)
    float* gx;         // bwd: grad_x (B,N,D)
    float s2;          // log2(e) / eps
    float t;           // coordinate pre-scale of the DIRECT form
    float inv_t;       // 1 / t
    float out_scale;   // -eps * ln(2)
    float clamp2;      // p = 1: floor on the (scaled) squared distance, 1e-8 * t^2 (utils.py:61)
    // fused Sinkhorn half-step (glhip_sinkhorn_step): h_j := h_j + pot_scale * pot_j ;  out_i := alpha * f_i + beta * prev_i
    const float* pot;  // (B,M) or NULL
    const float* prev; // (B,N) or NULL
    float pot_scale;   // 1 / eps
    float alpha, beta; // 1, 0 for the plain soft-min
    // gradient kernels in "value and gradient" mode (glhip_softmin_fwd_grad): `fwd` is only a GUESS of the soft-min, known to lie
    // within shift2 (base-2 units of LSE2) below the truth's upper bound; the kernel then also writes the exact value to `out`
    float shift2;      // 0 for the plain gradient
};

// dual vector entry j as the kernels see it (natural-log units)
template <typename T>
__device__ __forceinline__ float dual_entry(const SoftminParams<T>& p, long idx) {
    float hj = p.h[idx];
    if (p.pot) hj = __builtin_fmaf(p.pot[idx], p.pot_scale, hj);
    return hj;
}
// final value written for row idx from its base-2 log-sum-exp
template <typename T>
__device__ __forceinline__ float finish_value(const SoftminParams<T>& p, long idx, float lse2) {
    float f = p.alpha * (p.out_scale * lse2);
    if (p.prev) f = __builtin_fmaf(p.beta, p.prev[idx], f);
    return f;
}

// ---- shared pieces ---------------------------------------------------------------------------

template <int D, typename T>
__device__ __forceinline__ void centre_of(const T* __restrict__ x, int b, int N, int row0, float (&c)[D]) {
    // row0 is workgroup-uniform, so this is a scalar (SMEM) load shared by the whole workgroup
    load_point<D, T>(x, (long)b * N + row0, c);
}

// online log-sum-exp update with kChunk fresh exponents
__device__ __forceinline__ void lse_update(float& m, float& s, const float (&u)[kChunk]) {
    float cm = u[0];
#pragma unroll
    for (int c = 1; c < kChunk; ++c) cm = fmaxf(cm, u[c]);
    const float mn = fmaxf(m, cm);
    s *= fast_exp2(m - mn);
    m = mn;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int c = 0; c < kChunk; c += 2) {
        s0 += fast_exp2(u[c] - mn);
        s1 += fast_exp2(u[c + 1] - mn);
    }
    s += s0 + s1;
}

// u for one (row, record) pair
template <int D, int P, bool DIRECT>
__device__ __forceinline__ float pair_exponent(const float (&a)[D], const Rec<D>& r, float clamp2) {
    if (!DIRECT) {
        float u = rec_tail<D>(r);
#pragma unroll
        for (int d = D - 1; d >= 0; --d) u = __builtin_fmaf(a[d], r.c[d], u);
        return u;
    } else if (P == 2) {
        float u = rec_tail<D>(r);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float df = a[d] - r.c[d];
            u = __builtin_fmaf(-df, df, u);
        }
        return u;
    } else {
        float d2 = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float df = a[d] - r.c[d];
            d2 = __builtin_fmaf(df, df, d2);
        }
        return rec_tail<D>(r) - fast_sqrt(fmaxf(d2, clamp2));
    }
}

// ---- hard C-transform (the eps -> 0 limit of the soft-min) -------------------------------------------------------
//   out_i = min_j [ C(x_i, y_j) - g_j ],  C = |x-y|^2 / 2 (P = 2) or sqrt(max(|x-y|^2, 1e-8)) (P = 1), on explicit differences,
// natural units.  The eps = 0 branch of the reference's softmin_sample (ot/_implementations/sample.py:156-166).

template <typename T>
struct CminParams {
    const T* x;        // (B,N,D)
    const T* y;        // (B,M,D)
    const float* g;    // (B,M)
    float* out;        // (B,N)
};

template <int D_, int P, int R, typename T>
struct HardMinOp {
    static constexpr int kDim = D_;
    static constexpr int kRows = R;
    static constexpr int kPartial = 1;
    using Params = CminParams<T>;
    struct RowState {
        float a[R][D_];
        float best[R];
    };
    static constexpr float kHuge = 3.0e38f;

    static __device__ __forceinline__ void load_centre(const Params& p, int b, int N, int row0, float (&c)[D_]) {
        centre_of<D_, T>(p.x, b, N, row0, c);
    }
    static __device__ __forceinline__ void init_rows(const Params& p, int b, int N, int row0, int row_end, int tid,
                                                     const float (&c)[D_], RowState& st) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = min(row0 + r * kBlock + tid, row_end - 1);
            float xi[D_];
            load_point<D_, T>(p.x, (long)b * N + i, xi);
#pragma unroll
            for (int d = 0; d < D_; ++d) st.a[r][d] = xi[d] - c[d];
            st.best[r] = kHuge;
        }
    }
    static __device__ __forceinline__ Rec<D_> make_record(const Params& p, int b, int M, int j, const float (&c)[D_]) {
        float yj[D_];
        load_point<D_, T>(p.y, (long)b * M + j, yj);
        Rec<D_> rec;
#pragma unroll
        for (int d = 0; d < D_; ++d) rec.c[d] = yj[d] - c[d];
        rec_tail<D_>(rec) = p.g[(long)b * M + j];
        if (D_ == 2) rec.c[3] = 0.f;
        return rec;
    }
    static __device__ __forceinline__ Rec<D_> neutral_record() {
        Rec<D_> rec;
#pragma unroll
        for (int d = 0; d < D_; ++d) rec.c[d] = 0.f;
        rec_tail<D_>(rec) = -kHuge;      // C - g = +huge: never the minimum
        if (D_ == 2) rec.c[3] = 0.f;
        return rec;
    }
    static __device__ __forceinline__ void consume(RowState& st, const Rec<D_>* __restrict__ recs) {
#pragma unroll
        for (int c = 0; c < kChunk; ++c) {
            const Rec<D_> rc = recs[c];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float d2 = 0.f;
#pragma unroll
                for (int d = 0; d < D_; ++d) {
                    const float df = st.a[r][d] - rc.c[d];
                    d2 = __builtin_fmaf(df, df, d2);
                }
                const float C = (P == 2) ? 0.5f * d2 : fast_sqrt(fmaxf(d2, 1e-8f));
                st.best[r] = fminf(st.best[r], C - rec_tail<D_>(rc));
            }
        }
    }
    static __device__ __forceinline__ float value(float best) { return best > 1.0e37f ? __builtin_inff() : best; }   // empty set
    static __device__ __forceinline__ void finish_rows(const Params& p, int b, int N, int row0, int row_end, int tid,
                                                       const float (&)[D_], RowState& st) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = row0 + r * kBlock + tid;
            if (i < row_end) p.out[(long)b * N + i] = value(st.best[r]);
        }
    }
    static __device__ __forceinline__ void store_partial(const RowState& st, int r, float* dst) { dst[0] = st.best[r]; }
    static __device__ __forceinline__ void merge_row(const Params& p, int b, int N, int i, const float (&)[D_],
                                                     const float* part, int ns, long stride) {
        float best = kHuge;
        for (int k = 0; k < ns; ++k) best = fminf(best, part[k * stride]);
        p.out[(long)b * N + i] = value(best);
    }
};

// ---- forward ----------------------------------------------------------------------------------

template <int D_, int P, bool DIRECT, int R, typename T>
struct SoftminFwdOp {
    static constexpr int kDim = D_;
    static constexpr int kRows = R;
    using Params = SoftminParams<T>;
    struct RowState {
        float a[R][D_];   // scaled, centred row coordinates
        float r[R];       // row constant of the expanded form (0 for DIRECT)
        float m[R], s[R];
        float clamp2;
    };

    static __device__ __forceinline__ void load_centre(const Params& p, int b, int N, int row0, float (&c)[D_]) {
        centre_of<D_, T>(p.x, b, N, row0, c);
    }

    static __device__ __forceinline__ void init_rows(const Params& p, int b, int N, int row0, int row_end,
                                                     int tid, const float (&c)[D_], RowState& st) {
        st.clamp2 = p.clamp2;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = min(row0 + r * kBlock + tid, row_end - 1);   // clamp: idle lanes redo a valid row
            float xi[D_];
            load_point<D_, T>(p.x, (long)b * N + i, xi);
            float n2 = 0.f;
#pragma unroll
            for (int d = 0; d < D_; ++d) {
                const float xt = xi[d] - c[d];
                n2 = __builtin_fmaf(xt, xt, n2);
                st.a[r][d] = xt * (DIRECT ? p.t : p.s2);
            }
            st.r[r] = DIRECT ? 0.f : -0.5f * p.s2 * n2;
            st.m[r] = kNegBig;
            st.s[r] = 0.f;
        }
    }

    static __device__ __forceinline__ Rec<D_> make_record(const Params& p, int b, int M, int j, const float (&c)[D_]) {
        float yj[D_];
        load_point<D_, T>(p.y, (long)b * M + j, yj);
        Rec<D_> rec;
        float n2 = 0.f;
#pragma unroll
        for (int d = 0; d < D_; ++d) {
            const float yt = yj[d] - c[d];
            n2 = __builtin_fmaf(yt, yt, n2);
            rec.c[d] = DIRECT ? yt * p.t : yt;
        }
        const float hj = dual_entry(p, (long)b * M + j) * kLog2e;
        rec_tail<D_>(rec) = DIRECT ? hj : __builtin_fmaf(-0.5f * p.s2, n2, hj);
        if (D_ == 2) rec.c[3] = 0.f;
        return rec;
    }

    static __device__ __forceinline__ Rec<D_> neutral_record() {
        Rec<D_> rec;
#pragma unroll
        for (int d = 0; d < D_; ++d) rec.c[d] = 0.f;
        rec_tail<D_>(rec) = kNegBig;
        if (D_ == 2) rec.c[3] = 0.f;
        return rec;
    }

    static __device__ __forceinline__ void consume(RowState& st, const Rec<D_>* __restrict__ recs) {
        Rec<D_> rc[kChunk];
#pragma unroll
        for (int c = 0; c < kChunk; ++c) rc[c] = recs[c];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float u[kChunk];
#pragma unroll
            for (int c = 0; c < kChunk; ++c) u[c] = pair_exponent<D_, P, DIRECT>(st.a[r], rc[c], st.clamp2);
            lse_update(st.m[r], st.s[r], u);
        }
    }

    static __device__ __forceinline__ void finish_rows(const Params& p, int b, int N, int row0, int row_end,
                                                       int tid, const float (&)[D_], RowState& st) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = row0 + r * kBlock + tid;
            if (i < row_end) {
                const float lse2 = st.r[r] + st.m[r] + fast_log2(st.s[r]);
                p.out[(long)b * N + i] = finish_value(p, (long)b * N + i, lse2);
            }
        }
    }

    // column splits: partial state of a row = (max exponent, sum of 2^(u - max)); both centre-independent
    static constexpr int kPartial = 2;
    static __device__ __forceinline__ void store_partial(const RowState& st, int r, float* dst) {
        dst[0] = st.r[r] + st.m[r];
        dst[1] = st.s[r];
    }
    static __device__ __forceinline__ void merge_row(const Params& p, int b, int N, int i, const float (&)[D_],
                                                     const float* part, int ns, long stride) {
        float m = kNegBig, s = 0.f;
        for (int k = 0; k < ns; ++k) {
            const float mk = part[k * stride], sk = part[k * stride + 1];
            const float mn = fmaxf(m, mk);
            s = s * fast_exp2(m - mn) + sk * fast_exp2(mk - mn);
            m = mn;
        }
        p.out[(long)b * N + i] = finish_value(p, (long)b * N + i, m + fast_log2(s));
    }
};

// ---- backward with respect to x ------------------------------------------------------------------
//   P_ij = 2^(u_ij - LSE2_i);  p = 2: grad_x_i = g_i (x_i - sum_j P_ij y_j / sum_j P_ij)
//                              p = 1: grad_x_i = g_i  sum_j P_ij (x_i - y_j)/|x_i - y_j| / sum_j P_ij

template <int D_, int P, bool DIRECT, int R, typename T>
struct SoftminBwdOp {
    static constexpr int kDim = D_;
    static constexpr int kRows = R;
    using Params = SoftminParams<T>;
    struct RowState {
        float a[R][D_];
        float l[R];        // LSE2_i minus the row constant r_i
        float acc[R][D_];
        float sw[R];
        float clamp2;
    };

    static __device__ __forceinline__ void load_centre(const Params& p, int b, int N, int row0, float (&c)[D_]) {
        centre_of<D_, T>(p.x, b, N, row0, c);
    }

    static __device__ __forceinline__ void init_rows(const Params& p, int b, int N, int row0, int row_end,
                                                     int tid, const float (&c)[D_], RowState& st) {
        st.clamp2 = p.clamp2;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = min(row0 + r * kBlock + tid, row_end - 1);
            float xi[D_];
            load_point<D_, T>(p.x, (long)b * N + i, xi);
            float n2 = 0.f;
#pragma unroll
            for (int d = 0; d < D_; ++d) {
                const float xt = xi[d] - c[d];
                n2 = __builtin_fmaf(xt, xt, n2);
                st.a[r][d] = xt * (DIRECT ? p.t : p.s2);
                st.acc[r][d] = 0.f;
            }
            const float lse2 = p.fwd[(long)b * N + i] / p.out_scale;   // out = out_scale * LSE2
            st.l[r] = DIRECT ? lse2 : lse2 + 0.5f * p.s2 * n2;          // LSE2 - r_i
            st.sw[r] = 0.f;
        }
    }

    static __device__ __forceinline__ Rec<D_> make_record(const Params& p, int b, int M, int j, const float (&c)[D_]) {
        return SoftminFwdOp<D_, P, DIRECT, R, T>::make_record(p, b, M, j, c);
    }
    static __device__ __forceinline__ Rec<D_> neutral_record() {
        return SoftminFwdOp<D_, P, DIRECT, R, T>::neutral_record();
    }

    static __device__ __forceinline__ void consume(RowState& st, const Rec<D_>* __restrict__ recs) {
        Rec<D_> rc[kChunk];
#pragma unroll
        for (int c = 0; c < kChunk; ++c) rc[c] = recs[c];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int c = 0; c < kChunk; ++c) {
                const float u = pair_exponent<D_, P, DIRECT>(st.a[r], rc[c], st.clamp2);
                const float w = fast_exp2(u - st.l[r]);
                st.sw[r] += w;
                if (P == 2) {
#pragma unroll
                    for (int d = 0; d < D_; ++d) st.acc[r][d] = __builtin_fmaf(w, rc[c].c[d], st.acc[r][d]);
                } else {
                    float df[D_];
                    float d2 = 0.f;
#pragma unroll
                    for (int d = 0; d < D_; ++d) {
                        df[d] = st.a[r][d] - rc[c].c[d];
                        d2 = __builtin_fmaf(df[d], df[d], d2);
                    }
                    const float wr = (d2 > st.clamp2) ? w * fast_rsq(d2) : 0.f;
#pragma unroll
                    for (int d = 0; d < D_; ++d) st.acc[r][d] = __builtin_fmaf(wr, df[d], st.acc[r][d]);
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
                const float gi = p.g[(long)b * N + i];
                const float inv = (st.sw[r] > 0.f) ? 1.0f / st.sw[r] : 0.f;
#pragma unroll
                for (int d = 0; d < D_; ++d) {
                    float v;
                    if (P == 2) {
                        // rows: a = xt * scale; records: yt (expanded) or yt * t (direct)
                        const float xt = st.a[r][d] * (DIRECT ? p.inv_t : 1.0f / p.s2);
                        const float ybar = st.acc[r][d] * inv * (DIRECT ? p.inv_t : 1.0f);
                        v = xt - ybar;
                    } else {
                        v = st.acc[r][d] * inv;
                    }
                    p.gx[((long)b * N + i) * D_ + d] = gi * v;
                }
            }
        }
    }

    // column splits: partial sums (acc[D], sw) are additive; all splits of a row pass share its centre
    static constexpr int kPartial = D_ + 1;
    static __device__ __forceinline__ void store_partial(const RowState& st, int r, float* dst) {
#pragma unroll
        for (int d = 0; d < D_; ++d) dst[d] = st.acc[r][d];
        dst[D_] = st.sw[r];
    }
    static __device__ __forceinline__ void merge_row(const Params& p, int b, int N, int i, const float (&c)[D_],
                                                     const float* part, int ns, long stride) {
        float acc[D_], sw = 0.f;
#pragma unroll
        for (int d = 0; d < D_; ++d) acc[d] = 0.f;
        for (int k = 0; k < ns; ++k) {
#pragma unroll
            for (int d = 0; d < D_; ++d) acc[d] += part[k * stride + d];
            sw += part[k * stride + D_];
        }
        float xi[D_];
        load_point<D_, T>(p.x, (long)b * N + i, xi);
        const float gi = p.g ? p.g[(long)b * N + i] : 1.f;
        const float inv = (sw > 0.f) ? 1.0f / sw : 0.f;
        if (p.out) p.out[(long)b * N + i] = p.fwd[(long)b * N + i] + p.out_scale * (p.shift2 + fast_log2(sw));   // value-and-gradient mode
#pragma unroll
        for (int d = 0; d < D_; ++d) {
            float v;
            if (P == 2) v = (xi[d] - c[d]) - acc[d] * inv * (DIRECT ? p.inv_t : 1.0f);
            else v = acc[d] * inv;
            p.gx[((long)b * N + i) * D_ + d] = gi * v;
        }
    }
};

}  // namespace glhip
