// glhip_lines.h — log-sum-exp "convolution" along the lines of a regular grid: the primitive behind the reference's
// separable soft-min on images and volumes (`softmin_grid`, _legacy/utils.py:190-279, and the image Sinkhorn /
// barycenter code built on it):
//     out[r, i] = log sum_j exp( h[r, j] - c(i, j) ),     c(i, j) = (step (i - j))^2   (p = 2)   or   step |i - j|   (p = 1)
// for R independent lines of N samples.  The cost of a D-dimensional squared distance is separable, so the soft-min of
// an N^D image is D passes of this kernel (one per axis, R = B K N^(D-1) lines each): N^(D+1) pair evaluations instead
// of N^(2D).  At image sizes (N <= 1024) a pass is 1e7-1e9 pairs: launch-sized work, so this is a plain VALU kernel —
// one thread per output sample, the line staged in LDS and read by broadcast — with an exact running maximum
// refreshed every 8 samples (the reference's inputs contain -10000 for empty pixels, `log_dens`, next to O(1) values).
//
// The backward kernel is the transposed weighted sum  grad_h[r, j] = sum_i g[r, i] exp(h[r, j] - c(i, j) - out[r, i])
// (weights <= 1 by construction, no running maximum).
#pragma once

#include "glhip_common.h"

namespace glhip {

constexpr int kLineMax = 4096;          // samples per line held in LDS (2 floats each)

template <int P>
__device__ __forceinline__ float line_cost(int i, int j, float step2) {   // base-2 units; i - j is exact, one rounding in d
    const float d = (float)(i - j) * step2;
    return P == 2 ? d * d : fabsf(d);
}

// one workgroup per line; grid-stride over lines
template <int P>
__global__ void __launch_bounds__(kBlock)
lse_lines_fwd_kernel(const float* __restrict__ h, float* __restrict__ out, long R, int N, float step2) {
    // step2: coordinate spacing such that c(i,j) * log2(e) = (step2 (i-j))^2 (p = 2) or step2 |i-j| (p = 1)
    __shared__ float hs[kLineMax];
    for (long r = blockIdx.x; r < R; r += gridDim.x) {
        const float* hr = h + r * N;
        __syncthreads();
        for (int j = threadIdx.x; j < N; j += kBlock) hs[j] = hr[j] * kLog2e;
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += kBlock) {
            float m = kNegBig, s = 0.f;
            int j = 0;
            for (; j + kChunk <= N; j += kChunk) {
                float u[kChunk];
                float mc = m;
#pragma unroll
                for (int k = 0; k < kChunk; ++k) {
                    u[k] = hs[j + k] - line_cost<P>(i, j + k, step2);
                    mc = fmaxf(mc, u[k]);
                }
                s *= fast_exp2(m - mc);
                m = mc;
#pragma unroll
                for (int k = 0; k < kChunk; ++k) s += fast_exp2(u[k] - m);
            }
            for (; j < N; ++j) {
                const float u = hs[j] - line_cost<P>(i, j, step2);
                const float mc = fmaxf(m, u);
                s = s * fast_exp2(m - mc) + fast_exp2(u - mc);
                m = mc;
            }
            out[r * N + i] = (m + fast_log2(s)) * kLn2;
        }
    }
}

// hard C-transform along the lines of a grid:  out[r, i] = max_j [ g[r, j] - c(i, j) ],  c = (step (i-j))^2 or step |i-j| in
// natural units — `C_transform`, _legacy/utils.py:116-182 (KeOps LazyTensor.max there).
template <int P>
__global__ void __launch_bounds__(kBlock)
max_lines_kernel(const float* __restrict__ g, float* __restrict__ out, long R, int N, float step) {
    __shared__ float gs[kLineMax];
    for (long r = blockIdx.x; r < R; r += gridDim.x) {
        __syncthreads();
        for (int j = threadIdx.x; j < N; j += kBlock) gs[j] = g[r * N + j];
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += kBlock) {
            float best = -3.0e38f;
            for (int j = 0; j < N; ++j) best = fmaxf(best, gs[j] - line_cost<P>(i, j, step));
            out[r * N + i] = best;
        }
    }
}

template <int P>
__global__ void __launch_bounds__(kBlock)
lse_lines_bwd_kernel(const float* __restrict__ h, const float* __restrict__ lse, const float* __restrict__ g,
                     float* __restrict__ gh, long R, int N, float step2) {
    __shared__ float ls[kLineMax];   // lse_i * log2(e)
    __shared__ float gs[kLineMax];   // g_i
    for (long r = blockIdx.x; r < R; r += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += kBlock) {
            ls[i] = lse[r * N + i] * kLog2e;
            gs[i] = g[r * N + i];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < N; j += kBlock) {
            const float hj = h[r * N + j] * kLog2e;
            float acc = 0.f;
            for (int i = 0; i < N; ++i)
                acc = __builtin_fmaf(gs[i], fast_exp2(hj - line_cost<P>(i, j, step2) - ls[i]), acc);
            gh[r * N + j] = acc;
        }
    }
}

}  // namespace glhip
