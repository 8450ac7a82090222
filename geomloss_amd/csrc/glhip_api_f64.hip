// glhip_api_f64.hip — C-ABI part 7: the four reductions of the hot path in DOUBLE precision.
//
// The reference's matrix-free backends keep the dtype of their inputs (`softmin_online_lazytensor`,
// _legacy/sinkhorn_samples.py:229-290; `kernel_online`, _legacy/kernel_samples.py:128-137): float64 clouds are reduced in float64
// by KeOps.  Rounds 1-3 cast them down to fp32 with a warning.  These kernels are the float64 path: soft-min forward and its
// row gradient (p = 1, 2), kernel product and its row gradient (gaussian / laplacian / energy), dense, batched or block-sparse,
// for 1 <= D <= 16 (`f64_kernel`) and, since round 5, any larger D (`f64_generic_kernel`).  No matrix cores (there is no fp64 MFMA shape that helps an exp-bound reduction), no expanded form (explicit
// differences: nothing to cancel), 1 to 64 threads per row (small problems spread a row over a wavefront), columns staged through
// LDS as (D + 1) doubles, a lazy running maximum per row, an inlined 20-instruction `exp` (round 6: see f64_kernel).  It exists for
// callers who need the digits — the reference's hypothesis suite draws float64 half the time, `geomloss.ot` users land here.
//
// Of the fused entry points the half-step has a float64 form (glhip_sinkhorn_step_f64, round 6: the float64 coarse level of big
// two-scale losses and float64 `ot.solve_sample` calls are bound by their launch count); the one-launch iteration and the one-pass
// value + gradient do not: the host composes (geomloss_amd/hip.py).  Semantics follow the fp32 kernels: the clamp sqrt(max(d^2, 1e-8)) of utils.py:61 for p = 1 /
// laplacian (on x / blur) / energy, zero direction at clamped pairs, rows of an empty column set give -eps * (-inf) ... exactly
// what `logsumexp` gives, rows outside every row block are left untouched.
#include <cmath>

#include "glhip_common.h"
#include "glhip_error.h"

namespace glhip {
namespace {

enum F64Mode { F64_SOFTMIN = 0, F64_SOFTMIN_GRAD = 1, F64_KCONV = 2, F64_KCONV_GRAD = 3 };

struct F64Params {
    const double* x;      // (B, N, D)
    const double* y;      // (B, M, D)
    const double* s;      // (B, M): dual vector h | weights v
    const double* fwd;    // (B, N): soft-min values (F64_SOFTMIN_GRAD)
    const double* g;      // (B, N): incoming gradient (the two gradient modes)
    double* out;          // (B, N) | (B, N, D)
    double scale;         // eps | blur
    int p;                // cost exponent (soft-min modes)
    int kind;             // GLHIP_GAUSSIAN | LAPLACIAN | ENERGY (kernel modes)
    // fused half-step (F64_SOFTMIN, glhip_sinkhorn_step_f64): s_j := s_j + pot_scale * pot_j,  out_i := alpha * softmin_i + beta * prev_i
    const double* pot = nullptr;      // (B, M) or NULL
    const double* prev = nullptr;     // (B, N) or NULL
    double pot_scale = 0.0, alpha = 1.0, beta = 0.0;
};

constexpr int kF64Block = 256;
constexpr int kF64Tile = 128;      // columns per LDS tile: (DMAX + 1) * 128 doubles = 17 KiB at DMAX = 16

// e^t in double precision without the library call: n = rint(t log2 e), r = t - n ln2 (two-constant reduction, exact to 2^-100),
// e^r by its Taylor polynomial of degree 13 on |r| <= 0.347 (truncation 3e-18), v_ldexp_f64.  ~20 instructions against ~45 for
// ocml's exp with its special cases; relative error < 3e-16 (tests/test_f64_gpu.py holds the reductions to 1e-12 of the C oracle).
// t <= -1100 (and -inf) give 0; NaN propagates; t = +inf is the caller's business (the soft-min rescales before).
__device__ __forceinline__ double exp_f64(double t) {
    t = (t < -1100.0) ? -1100.0 : t;
    const double n = rint(t * 1.4426950408889634074);
    double r = fma(-n, 6.93147180369123816490e-01, t);       // ln2_hi (the 33 leading bits: n * ln2_hi is exact)
    r = fma(-n, 1.90821492927058770002e-10, r);              // ln2_lo
    double p = 1.0 / 6227020800.0;                            // 1 / 13!
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// TPR threads per row (a power of two <= 64: the threads of a row are lanes of one wavefront).  DMAX = 4 | 16: register arrays of DMAX
// doubles, loops fully unrolled and predicated on d < D.  Thread `sub` of a row takes the columns sub, sub + TPR, ... of every tile
// and the partial results meet in a butterfly at the end: 2 000 rows — the coarse level of a two-scale loss, the few hundred points
// of an `ot.solve_sample` call in float64 — fill the chip with TPR = 64 instead of sitting on 8 of its 256 CUs (round 6).
// Soft-min: a LAZY running maximum — doubles have the range for it: terms are summed against the maximum known so far and the sum
// is rescaled only when a term exceeds it by e^500 — so the inner loop has one exponential per pair and no divergent branch to speak of.
template <int MODE, int DMAX, int TPR>
__global__ void __launch_bounds__(kF64Block)
f64_kernel(F64Params prm, Ranges rg, int n_ranges, int N, int M, int D) {
    __shared__ double tile[kF64Tile * (DMAX + 1)];      // [column][D coordinates, scalar]
    constexpr int kRows = kF64Block / TPR;
    const int tid = threadIdx.x;
    const int r_in = tid / TPR, sub = tid % TPR;
    const int b = blockIdx.y;
    const bool sparse = n_ranges > 0;
    int row_begin, row_end, q_begin = 0, q_end = 1;
    if (sparse) {
        const int k = blockIdx.x;
        row_begin = rg.ranges_i[2 * k];
        row_end = rg.ranges_i[2 * k + 1];
        q_begin = k ? rg.slices_i[k - 1] : 0;
        q_end = rg.slices_i[k];
    } else {
        row_begin = blockIdx.x * kRows;
        row_end = min(N, row_begin + kRows);
    }
    const double* xb = prm.x + (long)b * N * D;
    const double* yb = prm.y + (long)b * M * D;
    const double* sb = prm.s + (long)b * M;
    const double inv_eps = 1.0 / prm.scale;
    const double inv_b2 = 1.0 / (prm.scale * prm.scale);

    for (int row0 = row_begin; row0 < row_end; row0 += kRows) {
        const int i = row0 + r_in;
        const bool live = i < row_end;
        double xi[DMAX];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) xi[d] = (live && d < D) ? xb[(long)i * D + d] : 0.0;
        double m = -INFINITY, ssum = 0.0, acc[DMAX];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) acc[d] = 0.0;
        const double fwd_i = (MODE == F64_SOFTMIN_GRAD && live) ? prm.fwd[(long)b * N + i] * inv_eps : 0.0;

        for (int q = q_begin; q < q_end; ++q) {
            const int js = sparse ? rg.redranges_j[2 * q] : 0, je = sparse ? rg.redranges_j[2 * q + 1] : M;
            for (int j0 = js; j0 < je; j0 += kF64Tile) {
                const int n = min(kF64Tile, je - j0);
                __syncthreads();
                for (int t = tid; t < n * (D + 1); t += kF64Block) {
                    const int c = t / (D + 1), d = t - c * (D + 1);
                    double v = (d < D) ? yb[(long)(j0 + c) * D + d] : sb[j0 + c];
                    if (MODE == F64_SOFTMIN && d == D && prm.pot) v = fma(prm.pot[(long)b * M + j0 + c], prm.pot_scale, v);
                    tile[c * (DMAX + 1) + d] = v;
                }
                __syncthreads();
                if (!live) continue;
                for (int c = sub; c < n; c += TPR) {
                    const double* rec = &tile[c * (DMAX + 1)];
                    double diff[DMAX], d2 = 0.0;
#pragma unroll
                    for (int d = 0; d < DMAX; ++d) {
                        diff[d] = (d < D) ? xi[d] - rec[d] : 0.0;
                        d2 = fma(diff[d], diff[d], d2);
                    }
                    const double sj = rec[D];
                    if (MODE == F64_SOFTMIN || MODE == F64_SOFTMIN_GRAD) {
                        const double cost = (prm.p == 2) ? 0.5 * d2 : sqrt(fmax(d2, 1e-8));
                        const double u = sj - cost * inv_eps;
                        if (MODE == F64_SOFTMIN) {
                            const double t = u - m;
                            if (t > 500.0) {                  // (also the first finite term: m = -inf) rescale, rare
                                ssum = ssum * exp_f64(m - u) + 1.0;
                                m = u;
                            } else if (u > -INFINITY) {       // (a massless column, h = -inf, adds nothing — also before any mass was seen)
                                ssum += exp_f64(t);
                            }
                        } else {
                            const double w = exp_f64(u + fwd_i);  // plan weight: sums to 1 over the row's columns
                            const double inv = (prm.p == 2) ? 1.0 : (d2 > 1e-8 ? 1.0 / sqrt(d2) : 0.0);
                            ssum += w;
#pragma unroll
                            for (int d = 0; d < DMAX; ++d) acc[d] = fma(w * inv, diff[d], acc[d]);
                        }
                    } else if (MODE == F64_KCONV) {
                        double k;
                        if (prm.kind == GLHIP_GAUSSIAN) k = exp_f64(-0.5 * d2 * inv_b2);
                        else if (prm.kind == GLHIP_LAPLACIAN) k = exp_f64(-sqrt(fmax(d2 * inv_b2, 1e-8)));
                        else k = -sqrt(fmax(d2, 1e-8));
                        ssum = fma(k, sj, ssum);
                    } else {
                        double coef;                          // d k / d x = coef * (x - y)
                        if (prm.kind == GLHIP_GAUSSIAN) coef = -exp_f64(-0.5 * d2 * inv_b2) * inv_b2;
                        else if (prm.kind == GLHIP_LAPLACIAN) {
                            const double dist = sqrt(d2);
                            coef = (d2 * inv_b2 > 1e-8) ? -exp_f64(-dist / prm.scale) / (prm.scale * dist) : 0.0;
                        } else coef = (d2 > 1e-8) ? -1.0 / sqrt(d2) : 0.0;
#pragma unroll
                        for (int d = 0; d < DMAX; ++d) acc[d] = fma(sj * coef, diff[d], acc[d]);
                    }
                }
            }
        }
        // the TPR partial results of a row: butterfly over its lanes (every lane of the wavefront takes part; dead rows carry neutral values)
        if constexpr (TPR > 1) {
#pragma unroll
            for (int off = TPR / 2; off > 0; off >>= 1) {
                if (MODE == F64_SOFTMIN) {
                    const double m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(ssum, off, 64);
                    const double mm = fmax(m, m2);
                    if (mm > -INFINITY) ssum = ssum * exp_f64(m - mm) + s2 * exp_f64(m2 - mm);
                    m = mm;
                } else {
                    ssum += __shfl_xor(ssum, off, 64);
#pragma unroll
                    for (int d = 0; d < DMAX; ++d) acc[d] += __shfl_xor(acc[d], off, 64);
                }
            }
        }
        if (!live || sub != 0) continue;
        const long idx = (long)b * N + i;
        if (MODE == F64_SOFTMIN) {
            double f = (m > -INFINITY) ? -prm.scale * (m + log(ssum)) : INFINITY;      // empty / massless row: -eps * (-inf)
            f *= prm.alpha;
            if (prm.prev) f = fma(prm.beta, prm.prev[idx], f);
            prm.out[idx] = f;
        } else if (MODE == F64_KCONV) {
            prm.out[idx] = ssum;
        } else {
            const double gi = prm.g[idx];
            const double norm = (MODE == F64_SOFTMIN_GRAD) ? (ssum > 0.0 ? gi / ssum : 0.0) : gi;
#pragma unroll
            for (int d = 0; d < DMAX; ++d)
                if (d < D) prm.out[idx * D + d] = norm * acc[d];
        }
    }
}

// D > 16: the same reductions with the coordinates looped at run time.  One thread per row; the row point is re-read from global
// memory (L1 / L2 hits) instead of living in registers, the columns are staged through a dynamic LDS tile of `tile_cols` records of
// (D + 1) doubles, and the gradient modes sweep the columns once per chunk of kF64Chunk coordinates so that their accumulators
// stay in registers (ceil(D / 16) sweeps).  Slower again than the D <= 16 kernel; it exists so that float64 clouds of ANY
// dimension keep their dtype, as they do in the reference (`Vi(D)` formulas, _legacy/sinkhorn_samples.py:322-334).
constexpr int kF64Chunk = 16;

template <int MODE>
__global__ void __launch_bounds__(kF64Block)
f64_generic_kernel(F64Params prm, Ranges rg, int n_ranges, int N, int M, int D, int tile_cols) {
    extern __shared__ double gtile[];      // [column][D coordinates, scalar]
    constexpr bool GRAD = (MODE == F64_SOFTMIN_GRAD || MODE == F64_KCONV_GRAD);
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const bool sparse = n_ranges > 0;
    int row_begin, row_end, q_begin = 0, q_end = 1;
    if (sparse) {
        const int k = blockIdx.x;
        row_begin = rg.ranges_i[2 * k];
        row_end = rg.ranges_i[2 * k + 1];
        q_begin = k ? rg.slices_i[k - 1] : 0;
        q_end = rg.slices_i[k];
    } else {
        row_begin = blockIdx.x * kF64Block;
        row_end = min(N, row_begin + kF64Block);
    }
    const double* xb = prm.x + (long)b * N * D;
    const double* yb = prm.y + (long)b * M * D;
    const double* sb = prm.s + (long)b * M;
    const double inv_eps = 1.0 / prm.scale;
    const double inv_b2 = 1.0 / (prm.scale * prm.scale);
    const int stride = D + 1;

    for (int row0 = row_begin; row0 < row_end; row0 += kF64Block) {
        const int i = row0 + tid;
        const bool live = i < row_end;
        const double* xi = xb + (long)(live ? i : row_begin) * D;
        const long idx = (long)b * N + i;
        const double fwd_i = (MODE == F64_SOFTMIN_GRAD && live) ? prm.fwd[idx] * inv_eps : 0.0;
        const int sweeps = GRAD ? (D + kF64Chunk - 1) / kF64Chunk : 1;
        for (int sweep = 0; sweep < sweeps; ++sweep) {
            const int d0 = sweep * kF64Chunk;
            double m = -INFINITY, ssum = 0.0, acc[kF64Chunk];
#pragma unroll
            for (int k = 0; k < kF64Chunk; ++k) acc[k] = 0.0;
            for (int q = q_begin; q < q_end; ++q) {
                const int js = sparse ? rg.redranges_j[2 * q] : 0, je = sparse ? rg.redranges_j[2 * q + 1] : M;
                for (int j0 = js; j0 < je; j0 += tile_cols) {
                    const int n = min(tile_cols, je - j0);
                    __syncthreads();
                    for (int t = tid; t < n * stride; t += kF64Block) {
                        const int c = t / stride, d = t - c * stride;
                        double v = (d < D) ? yb[(long)(j0 + c) * D + d] : sb[j0 + c];
                        if (MODE == F64_SOFTMIN && d == D && prm.pot) v = fma(prm.pot[(long)b * M + j0 + c], prm.pot_scale, v);
                        gtile[t] = v;
                    }
                    __syncthreads();
                    if (!live) continue;
                    for (int c = 0; c < n; ++c) {
                        const double* rec = &gtile[c * stride];
                        double d2 = 0.0;
                        for (int d = 0; d < D; ++d) {
                            const double df = xi[d] - rec[d];
                            d2 = fma(df, df, d2);
                        }
                        const double sj = rec[D];
                        double coef = 0.0;            // gradient modes: this pair adds coef * (x - y) to the row
                        if (MODE == F64_SOFTMIN) {
                            const double cost = (prm.p == 2) ? 0.5 * d2 : sqrt(fmax(d2, 1e-8));
                            const double u = sj - cost * inv_eps;
                            if (u > m) {
                                ssum = ssum * exp(m - u) + 1.0;
                                m = u;
                            } else if (u > -INFINITY) {
                                ssum += exp(u - m);
                            }
                        } else if (MODE == F64_SOFTMIN_GRAD) {
                            const double cost = (prm.p == 2) ? 0.5 * d2 : sqrt(fmax(d2, 1e-8));
                            const double w = exp(sj - cost * inv_eps + fwd_i);
                            ssum += w;
                            coef = w * ((prm.p == 2) ? 1.0 : (d2 > 1e-8 ? 1.0 / sqrt(d2) : 0.0));
                        } else if (MODE == F64_KCONV) {
                            double k;
                            if (prm.kind == GLHIP_GAUSSIAN) k = exp(-0.5 * d2 * inv_b2);
                            else if (prm.kind == GLHIP_LAPLACIAN) k = exp(-sqrt(fmax(d2 * inv_b2, 1e-8)));
                            else k = -sqrt(fmax(d2, 1e-8));
                            ssum = fma(k, sj, ssum);
                        } else {
                            if (prm.kind == GLHIP_GAUSSIAN) coef = -exp(-0.5 * d2 * inv_b2) * inv_b2;
                            else if (prm.kind == GLHIP_LAPLACIAN) {
                                const double dist = sqrt(d2);
                                coef = (d2 * inv_b2 > 1e-8) ? -exp(-dist / prm.scale) / (prm.scale * dist) : 0.0;
                            } else coef = (d2 > 1e-8) ? -1.0 / sqrt(d2) : 0.0;
                            coef *= sj;
                        }
                        if (GRAD) {
#pragma unroll
                            for (int k = 0; k < kF64Chunk; ++k)
                                if (d0 + k < D) acc[k] = fma(coef, xi[d0 + k] - rec[d0 + k], acc[k]);
                        }
                    }
                }
            }
            if (!live) continue;
            if (MODE == F64_SOFTMIN) {
                double f = (m > -INFINITY) ? -prm.scale * (m + log(ssum)) : INFINITY;
                f *= prm.alpha;
                if (prm.prev) f = fma(prm.beta, prm.prev[idx], f);
                prm.out[idx] = f;
            } else if (MODE == F64_KCONV) {
                prm.out[idx] = ssum;
            } else {
                const double gi = prm.g[idx];
                const double norm = (MODE == F64_SOFTMIN_GRAD) ? (ssum > 0.0 ? gi / ssum : 0.0) : gi;
#pragma unroll
                for (int k = 0; k < kF64Chunk; ++k)
                    if (d0 + k < D) prm.out[idx * D + d0 + k] = norm * acc[k];
            }
        }
    }
}

template <int MODE>
int launch_f64(const char* fn, const F64Params& prm, const int32_t* ri, const int32_t* si, const int32_t* rj, int n_ranges, int B, int N,
               int M, int D, void* stream) {
    if (B < 0 || N < 0 || M < 0 || D < 1) return fail(GLHIP_EINVAL, "%s: bad sizes B=%d N=%d M=%d D=%d", fn, B, N, M, D);
    if (D > 4095) return fail(GLHIP_EUNSUPPORTED, "%s: the float64 kernels serve D <= 4095 (got %d)", fn, D);      // one LDS record per column
    if (n_ranges < 0) return fail(GLHIP_EINVAL, "%s: n_ranges < 0", fn);
    if (n_ranges > 0 && (!ri || !si || !rj)) return fail(GLHIP_EINVAL, "%s: block-sparse mode needs ranges_i, slices_i, redranges_j", fn);
    if (n_ranges > 0 && B != 1) return fail(GLHIP_EUNSUPPORTED, "%s: block-sparse mode requires B == 1 (got %d)", fn, B);
    if (B > 65535) return fail(GLHIP_EUNSUPPORTED, "%s: B=%d exceeds the grid.y limit 65535", fn, B);
    if (B == 0 || N == 0) return GLHIP_OK;
    if (!prm.x || !prm.out || ((!prm.y || !prm.s) && M > 0)) return fail(GLHIP_EINVAL, "%s: NULL pointer", fn);
    const Ranges rg{ri, si, rj, nullptr};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (D <= 16) {
        // threads per row: enough of them to put ~4 wavefronts on every SIMD (256 CUs x 4 SIMDs x 4 x 64 lanes), while every thread
        // keeps at least 4 columns of a 128-column tile; block-sparse launches: also no wider than their row blocks are tall
        const long rows = (long)B * N;
        int tpr = 1;
        while (tpr < 64 && rows * tpr < 262144 && (long)M >= 8L * tpr) tpr *= 4;
        if (n_ranges > 0) while (tpr > 1 && (long)(kF64Block / tpr) * n_ranges < N) tpr /= 4;
#define GL_F64(DM, T) hipLaunchKernelGGL((f64_kernel<MODE, DM, T>), dim3(n_ranges > 0 ? n_ranges : (N + kF64Block / T - 1) / (kF64Block / T), B, 1), \
                                         dim3(kF64Block), 0, st, prm, rg, n_ranges, N, M, D)
        if (D <= 4) { if (tpr == 1) GL_F64(4, 1); else if (tpr == 4) GL_F64(4, 4); else if (tpr == 16) GL_F64(4, 16); else GL_F64(4, 64); }
        else { if (tpr == 1) GL_F64(16, 1); else if (tpr == 4) GL_F64(16, 4); else if (tpr == 16) GL_F64(16, 16); else GL_F64(16, 64); }
#undef GL_F64
    } else {
        const dim3 grid(n_ranges > 0 ? n_ranges : (N + kF64Block - 1) / kF64Block, B, 1);
        int tile_cols = (int)(32768 / ((size_t)(D + 1) * sizeof(double)));      // <= 32 KiB of LDS per workgroup
        tile_cols = tile_cols < 1 ? 1 : (tile_cols > kF64Tile ? kF64Tile : tile_cols);
        const size_t lds = (size_t)tile_cols * (D + 1) * sizeof(double);
        hipLaunchKernelGGL((f64_generic_kernel<MODE>), grid, dim3(kF64Block), lds, st, prm, rg, n_ranges, N, M, D, tile_cols);
    }
    return check_launch(fn);
}

}  // namespace
}  // namespace glhip

using namespace glhip;

extern "C" {

int glhip_softmin_fwd_f64(const double* x, const double* y, const double* h, double* out, int B, int N, int M, int D, double eps, int p,
                          const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges, void* stream) {
    if (!(eps > 0.0)) return fail(GLHIP_EINVAL, "glhip_softmin_fwd_f64: eps must be > 0");
    if (p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "glhip_softmin_fwd_f64: p must be 1 or 2 (got %d)", p);
    const F64Params prm{x, y, h, nullptr, nullptr, out, eps, p, 0};
    return launch_f64<F64_SOFTMIN>("glhip_softmin_fwd_f64", prm, ranges_i, slices_i, redranges_j, n_ranges, B, N, M, D, stream);
}

int glhip_sinkhorn_step_f64(const double* x, const double* y, const double* logw, const double* pot, const double* prev, double* out,
                            int B, int N, int M, int D, double eps, double damping, int p, const int32_t* ranges_i,
                            const int32_t* slices_i, const int32_t* redranges_j, int n_ranges, void* stream) {
    if (!(eps > 0.0)) return fail(GLHIP_EINVAL, "glhip_sinkhorn_step_f64: eps must be > 0");
    if (p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "glhip_sinkhorn_step_f64: p must be 1 or 2 (got %d)", p);
    if (prev && prev == out) return fail(GLHIP_EINVAL, "glhip_sinkhorn_step_f64: out must not alias prev");
    F64Params prm{x, y, logw, nullptr, nullptr, out, eps, p, 0};
    prm.pot = pot;
    prm.pot_scale = 1.0 / eps;
    prm.prev = prev;
    prm.alpha = prev ? 0.5 * damping : damping;
    prm.beta = 0.5;
    return launch_f64<F64_SOFTMIN>("glhip_sinkhorn_step_f64", prm, ranges_i, slices_i, redranges_j, n_ranges, B, N, M, D, stream);
}

int glhip_softmin_bwd_x_f64(const double* x, const double* y, const double* h, const double* out, const double* grad_out, double* grad_x,
                            int B, int N, int M, int D, double eps, int p, const int32_t* ranges_i, const int32_t* slices_i,
                            const int32_t* redranges_j, int n_ranges, void* stream) {
    if (!(eps > 0.0)) return fail(GLHIP_EINVAL, "glhip_softmin_bwd_x_f64: eps must be > 0");
    if (p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "glhip_softmin_bwd_x_f64: p must be 1 or 2 (got %d)", p);
    if ((long)B * N > 0 && (!out || !grad_out)) return fail(GLHIP_EINVAL, "glhip_softmin_bwd_x_f64: NULL out / grad_out");
    const F64Params prm{x, y, h, out, grad_out, grad_x, eps, p, 0};
    return launch_f64<F64_SOFTMIN_GRAD>("glhip_softmin_bwd_x_f64", prm, ranges_i, slices_i, redranges_j, n_ranges, B, N, M, D, stream);
}

int glhip_kernel_conv_fwd_f64(int kind, const double* x, const double* y, const double* v, double* out, int B, int N, int M, int D,
                              double blur, const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges,
                              void* stream) {
    if (kind < GLHIP_GAUSSIAN || kind > GLHIP_ENERGY) return fail(GLHIP_EINVAL, "glhip_kernel_conv_fwd_f64: bad kind %d", kind);
    if (kind != GLHIP_ENERGY && !(blur > 0.0)) return fail(GLHIP_EINVAL, "glhip_kernel_conv_fwd_f64: blur must be > 0");
    const F64Params prm{x, y, v, nullptr, nullptr, out, kind == GLHIP_ENERGY ? 1.0 : blur, 2, kind};
    return launch_f64<F64_KCONV>("glhip_kernel_conv_fwd_f64", prm, ranges_i, slices_i, redranges_j, n_ranges, B, N, M, D, stream);
}

int glhip_kernel_conv_bwd_x_f64(int kind, const double* x, const double* y, const double* v, const double* grad_out, double* grad_x, int B,
                                int N, int M, int D, double blur, const int32_t* ranges_i, const int32_t* slices_i,
                                const int32_t* redranges_j, int n_ranges, void* stream) {
    if (kind < GLHIP_GAUSSIAN || kind > GLHIP_ENERGY) return fail(GLHIP_EINVAL, "glhip_kernel_conv_bwd_x_f64: bad kind %d", kind);
    if (kind != GLHIP_ENERGY && !(blur > 0.0)) return fail(GLHIP_EINVAL, "glhip_kernel_conv_bwd_x_f64: blur must be > 0");
    if ((long)B * N > 0 && !grad_out) return fail(GLHIP_EINVAL, "glhip_kernel_conv_bwd_x_f64: NULL grad_out");
    const F64Params prm{x, y, v, nullptr, grad_out, grad_x, kind == GLHIP_ENERGY ? 1.0 : blur, 2, kind};
    return launch_f64<F64_KCONV_GRAD>("glhip_kernel_conv_bwd_x_f64", prm, ranges_i, slices_i, redranges_j, n_ranges, B, N, M, D, stream);
}

}  // extern "C"
