// glhip_api_f64.hip — C-ABI part 7: the four reductions of the hot path in DOUBLE precision.
//
// The reference's matrix-free backends keep the dtype of their inputs (`softmin_online_lazytensor`,
// _legacy/sinkhorn_samples.py:229-290; `kernel_online`, _legacy/kernel_samples.py:128-137): float64 clouds are reduced in float64
// by KeOps.  Rounds 1-3 cast them down to fp32 with a warning.  These kernels are the float64 path: soft-min forward and its
// row gradient (p = 1, 2), kernel product and its row gradient (gaussian / laplacian / energy), dense, batched or block-sparse,
// for 1 <= D <= 16 (`f64_kernel`) and, since round 5, any larger D (`f64_generic_kernel`).  No matrix cores (there is no fp64 MFMA shape that helps an exp-bound reduction), no expanded form (explicit
// differences: nothing to cancel), one thread per row, columns staged through LDS as (D + 1) doubles, a running maximum per
// row.  MI355X retires a float64 `exp` in ~40 VALU instructions, so this path runs at ~4-5e11 pairs/s — 20-25x below the fp32
// kernels, which is what float64 costs on this part; it exists for callers who need the digits, not the speed.
//
// The fused entry points (half-step, one-launch iteration, one-pass value + gradient) have no float64 form: the host composes
// (geomloss_amd/hip.py).  Semantics follow the fp32 kernels: the clamp sqrt(max(d^2, 1e-8)) of utils.py:61 for p = 1 /
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
};

constexpr int kF64Block = 256;
constexpr int kF64Tile = 128;      // columns per LDS tile: (DMAX + 1) * 128 doubles = 17 KiB at DMAX = 16

// One thread per row.  DMAX = 4 | 16: register arrays of DMAX doubles, loops fully unrolled and predicated on d < D.
template <int MODE, int DMAX>
__global__ void __launch_bounds__(kF64Block)
f64_kernel(F64Params prm, Ranges rg, int n_ranges, int N, int M, int D) {
    __shared__ double tile[kF64Tile * (DMAX + 1)];      // [column][D coordinates, scalar]
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

    for (int row0 = row_begin; row0 < row_end; row0 += kF64Block) {
        const int i = row0 + tid;
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
                    tile[c * (DMAX + 1) + d] = (d < D) ? yb[(long)(j0 + c) * D + d] : sb[j0 + c];
                }
                __syncthreads();
                if (!live) continue;
                for (int c = 0; c < n; ++c) {
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
                            if (u > m) {                      // new maximum: rescale the running sum
                                ssum = ssum * exp(m - u) + 1.0;
                                m = u;
                            } else if (u > -INFINITY) {
                                ssum += exp(u - m);
                            }
                        } else {
                            const double w = exp(u + fwd_i);  // plan weight: sums to 1 over the row's columns
                            const double inv = (prm.p == 2) ? 1.0 : (d2 > 1e-8 ? 1.0 / sqrt(d2) : 0.0);
                            ssum += w;
#pragma unroll
                            for (int d = 0; d < DMAX; ++d) acc[d] = fma(w * inv, diff[d], acc[d]);
                        }
                    } else if (MODE == F64_KCONV) {
                        double k;
                        if (prm.kind == GLHIP_GAUSSIAN) k = exp(-0.5 * d2 * inv_b2);
                        else if (prm.kind == GLHIP_LAPLACIAN) k = exp(-sqrt(fmax(d2 * inv_b2, 1e-8)));
                        else k = -sqrt(fmax(d2, 1e-8));
                        ssum = fma(k, sj, ssum);
                    } else {
                        double coef;                          // d k / d x = coef * (x - y)
                        if (prm.kind == GLHIP_GAUSSIAN) coef = -exp(-0.5 * d2 * inv_b2) * inv_b2;
                        else if (prm.kind == GLHIP_LAPLACIAN) {
                            const double dist = sqrt(d2);
                            coef = (d2 * inv_b2 > 1e-8) ? -exp(-dist / prm.scale) / (prm.scale * dist) : 0.0;
                        } else coef = (d2 > 1e-8) ? -1.0 / sqrt(d2) : 0.0;
#pragma unroll
                        for (int d = 0; d < DMAX; ++d) acc[d] = fma(sj * coef, diff[d], acc[d]);
                    }
                }
            }
        }
        if (!live) continue;
        const long idx = (long)b * N + i;
        if (MODE == F64_SOFTMIN) {
            prm.out[idx] = (m > -INFINITY) ? -prm.scale * (m + log(ssum)) : INFINITY;      // empty / massless row: -eps * (-inf)
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
                        gtile[t] = (d < D) ? yb[(long)(j0 + c) * D + d] : sb[j0 + c];
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
                prm.out[idx] = (m > -INFINITY) ? -prm.scale * (m + log(ssum)) : INFINITY;
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
    const dim3 grid(n_ranges > 0 ? n_ranges : (N + kF64Block - 1) / kF64Block, B, 1);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (D <= 4) hipLaunchKernelGGL((f64_kernel<MODE, 4>), grid, dim3(kF64Block), 0, st, prm, rg, n_ranges, N, M, D);
    else if (D <= 16) hipLaunchKernelGGL((f64_kernel<MODE, 16>), grid, dim3(kF64Block), 0, st, prm, rg, n_ranges, N, M, D);
    else {
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
