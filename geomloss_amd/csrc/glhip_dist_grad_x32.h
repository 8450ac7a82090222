// glhip_dist_grad_x32.h — laplacian / energy kernel products TOGETHER WITH their gradient in the row points (or the gradient alone),
// with the squared distance on the matrix cores: the gradient counterpart of glhip_dist_x32.h.
//
//     out_i      = sum_j k(x_i, y_j) v_j                     k = exp(-|x-y| / blur)  |  -|x-y|          (kernel_samples.py:71-82)
//     unit_i[d]  = d out_i / d x_i[d] = gscale * sum_j v_j k'_ij (xs_i - ys_j)[d] / |xs_i - ys_j|        (scaled coordinates xs = t (x - c))
//                = gscale * ( xs_i[d] S0_i - S1_i[d] ),    S0 = sum_j w_ij,  S1 = sum_j w_ij ys_j,  w_ij = v_j k'_ij / |xs_i - ys_j|
// with k' = k (laplacian) or 1 (energy).  The direct-difference operators (ConvOp MODE 1 / 2, glhip_kconv_ops.h) spend 9-13 VALU
// instructions per pair on this; here the 32 x 32 block of squared distances comes out of the chained MFMA pair of the product
// kernel, the lane owns ONE row (so the accumulators are 1 + 1 + D registers), and per pair the VALU runs
//     v_rsq_f32, v_mul (|.| = d2 rsq(d2)), [v_exp_f32, v_mul: laplacian], fma (product), fma (S0), D x fma (S1)
// reading (v_j, v_j ys_j) of 4 consecutive columns as broadcast float4 from LDS.  |.| = m rsq(m) is the arithmetic of the fused
// VALU mode too: the other terms of the same kernel norm are sent to the product kernel of glhip_dist_x32.h in its FAMILY
// variant (same expression), so that the rounding bias of the three terms stays common (GLHIP_FLAG_GRAD_FAMILY, glhip.h).
//
// Same conditions and the same near-pair machinery as the product kernel (block-sparse launches over spatially compact row blocks;
// pairs below thr = max(guard |xs_i|^2, clamp2) re-evaluated on explicit differences).  Pairs inside the clamp of utils.py:61 add
// k(clamp) v_j to the product and nothing to the gradient — what the reference's `sqrt(clamp_min(., 1e-8))` gives under autograd.
#pragma once

#include "glhip_dist_x32.h"

namespace glhip {

enum DistGradMode { DG_FWDGRAD = 0, DG_BWD = 1 };   // product + unit gradient | gradient scaled by the incoming g_i

template <typename T>
struct DistGradParams {
    DistParams<T> d;      // x, y, s = v, out (DG_FWDGRAD), t, clamp2, guard
    const float* g;       // DG_BWD: (N)
    float* gx;            // (N, D)
    float gscale;         // -1 / blur (laplacian), -1 (energy)
};

// one 32 x 32 block: d2 (scaled squared distances, near pairs already exact and floored at clamp2), q = &wq[first column of the group
// + 4 * half] with wq[c][column] = (v, v ys_0, v ys_1, v ys_2)[c].  GUARDED: pairs with d2 <= clamp2 get no direction.
template <int KIND, int D, bool GUARDED, bool PRODUCT>
__device__ __forceinline__ void dist_grad_block(const f32x16& d2, const float* __restrict__ q, int qstride, float clamp2, float& accP,
                                                float& S0, float (&S1)[3]) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const float4 v4 = *reinterpret_cast<const float4*>(q + qq * 8);
        float4 y4[3];
#pragma unroll
        for (int d = 0; d < D; ++d) y4[d] = *reinterpret_cast<const float4*>(q + (1 + d) * qstride + qq * 8);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m = __builtin_fabsf(d2[qq * 4 + r]);
            const float rs = fast_rsq(m);
            const float dist = m * rs;                              // |xs_i - ys_j| (FAMILY arithmetic)
            float w;                                                // k' / |.|
            if (KIND == GLHIP_LAPLACIAN) {
                const float k = fast_exp2(-dist);
                if (PRODUCT) accP = __builtin_fmaf(k, vv[r], accP);
                w = k * rs;
            } else {
                if (PRODUCT) accP = __builtin_fmaf(-dist, vv[r], accP);
                w = rs;
            }
            if (GUARDED) w = (m > clamp2) ? w : 0.f;
            S0 = __builtin_fmaf(w, vv[r], S0);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float yv = (r == 0) ? y4[d].x : (r == 1) ? y4[d].y : (r == 2) ? y4[d].z : y4[d].w;
                S1[d] = __builtin_fmaf(w, yv, S1[d]);
            }
        }
    }
}

template <int KIND, int GM, int D, typename T, int NW>
__global__ void __launch_bounds__(NW * 64, 6)       // <= 80 VGPRs: three 8-wavefront workgroups per CU (41 KiB of LDS each)
dist_grad_x32_kernel(DistGradParams<T> gp, Ranges rg, int N, int M, SplitInfo sp) {
    static_assert(KIND == GLHIP_LAPLACIAN || KIND == GLHIP_ENERGY, "distance-type kernels");
    constexpr int kRowsPerBlock = NW * 32;
    constexpr int kThreads = NW * 64;
    constexpr bool PRODUCT = GM == DG_FWDGRAD;
    constexpr int kPart = PRODUCT ? D + 1 : D;        // ConvOp<KIND, D, 1, T, 2 | 1>::kPartial
    __shared__ uint4 tile[kDistTile * 4];                                         // [column group of 32][K block 0..3][column]
    __shared__ __attribute__((aligned(16))) float wq[(1 + D) * kDistTile];        // [v | v ys_d][column]
    __shared__ float csum[NW][4];
    const DistParams<T>& prm = gp.d;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.z;
    const int ns = sp.n_splits;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int rec0 = half * 32 + l31;

    int row_begin, row_end, q_begin, q_end;
    block_extent<true>(rg, N, kRowsPerBlock, row_begin, row_end, q_begin, q_end, blockIdx.x);

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint4 kZero = uint4{0u, 0u, 0u, 0u};

    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerBlock) {
        float centre[D];       // mean of the rows of the pass (see glhip_dist_x32.h)
        {
            const int cnt = min(row_end, row0 + kRowsPerBlock) - row0;
            float part[D];
#pragma unroll
            for (int d = 0; d < D; ++d) part[d] = 0.f;
            if (tid < cnt) {
                float xi[D];
                load_point<D, T>(prm.x, row0 + tid, xi);
#pragma unroll
                for (int d = 0; d < D; ++d) part[d] = xi[d];
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                for (int off = 32; off > 0; off >>= 1) part[d] += __shfl_xor(part[d], off, 64);
            }
            __syncthreads();
            if (lane == 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) csum[wave][d] = part[d];
            }
            __syncthreads();
#pragma unroll
            for (int d = 0; d < D; ++d) {
                float tot = 0.f;
                for (int w = 0; w < NW; ++w) tot += csum[w][d];
                centre[d] = tot / (float)cnt;
            }
        }

        const int wave_row0 = row0 + wave * 32;
        const bool wave_active = wave_row0 < row_end;
        uint4 Xlo, Xhi;
        float xs3[3] = {0.f, 0.f, 0.f}, thr, g2;
        {
            const int i = min(wave_row0 + l31, row_end - 1);
            float xi[D];
            load_point<D, T>(prm.x, i, xi);
            float a[3] = {0.f, 0.f, 0.f}, n2 = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float xs = (xi[d] - centre[d]) * prm.t;
                xs3[d] = xs;
                n2 = __builtin_fmaf(xs, xs, n2);
                a[d] = -2.f * xs;
            }
            g2 = prm.guard * n2;
            thr = fmaxf(g2, prm.clamp2);
            const uint4 p0 = pack_a(a[0]), p1 = (D > 1) ? pack_a(a[1]) : kZero, p2 = (D > 2) ? pack_a(a[2]) : kZero;
            Xlo = select_u4(half != 0, p1, p0);
            Xhi = select_u4(half != 0, pack_negmax(-n2), p2);
        }
        float near2;
        {
            float r2 = wave_active ? g2 : 0.f;
            for (int off = 32; off > 0; off >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, off, 64));
            __syncthreads();
            if (lane == 0) csum[wave][3] = r2;
            __syncthreads();
            float tot = 0.f;
            for (int w = 0; w < NW; ++w) tot = fmaxf(tot, csum[w][3]);
            const float rmax = fast_sqrt(tot / fmaxf(prm.guard, 1e-30f)), reach = 1.07f * rmax + 2.f * fast_sqrt(prm.clamp2);
            near2 = (prm.guard > 0.f) ? reach * reach : 3.0e38f;      // guard off: every tile takes the guarded (clamp-aware) path
        }
        float accP = 0.f, S0 = 0.f, S1[3] = {0.f, 0.f, 0.f};

        for (int q = q_begin + split; q < q_end; q += ns) {
            const int js = rg.redranges_j[2 * q], je = rg.redranges_j[2 * q + 1];
            for (int j0 = js; j0 < je; j0 += kDistTile) {
                const int n = min(kDistTile, je - j0);
                const int npad = (n + 31) & ~31;
                __syncthreads();
                int near = 0;
                for (int t = tid; t < npad; t += kThreads) {
                    float ys[3] = {0.f, 0.f, 0.f}, n2 = 0.f, vj = 0.f;
                    if (t < n) {
                        float yj[D];
                        load_point<D, T>(prm.y, j0 + t, yj);
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            ys[d] = (yj[d] - centre[d]) * prm.t;
                            n2 = __builtin_fmaf(ys[d], ys[d], n2);
                        }
                        vj = prm.s[j0 + t];
                    } else {
                        n2 = 1.0e30f;           // padding: infinitely far, zero weight
                    }
                    uint4* base = &tile[(t >> 5) * 128 + (t & 31)];
#pragma unroll
                    for (int d = 0; d < 3; ++d) base[d * 32] = (d < D) ? pack_y(ys[d]) : kZero;
                    base[3 * 32] = pack_h1(n2);
                    wq[t] = vj;
#pragma unroll
                    for (int d = 0; d < D; ++d) wq[(1 + d) * kDistTile + t] = vj * ys[d];
                    near |= (t < n && n2 < near2) ? 1 : 0;
                }
                const bool tile_near = __syncthreads_or(near) != 0;
                if (!wave_active) continue;

                // Sums are hierarchical — block (16 pairs per lane) -> tile -> pass — like those of the product kernel: a single
                // running fp32 accumulator per row silently absorbs the many far pairs whose term is below half an ulp of the sum
                // (laplacian at 3e5 points: -1.7e-5 on a product, the same in the direct-difference operators), and a kernel norm is
                // a difference of three such products: its terms must lose the same mass (profiles/r03_family_bias.txt).
                float tP = 0.f, t0 = 0.f, t1[3] = {0.f, 0.f, 0.f};
                auto loop = [&](auto guarded) {
                    constexpr bool GD = decltype(guarded)::value;
                    for (int G = 0; G < npad / 32; ++G) {
                        const uint4* g = &tile[G * 128];
                        f32x16 d2 = mfma_x32(g[rec0], Xlo, zero16);
                        d2 = mfma_x32(g[64 + rec0], Xhi, d2);
                        if constexpr (GD) {
                            if (__any(min16(d2) < thr)) exact_near_pairs<D>(d2, thr, prm.clamp2, g, xs3, half);
                        }
                        float bP = 0.f, b0 = 0.f, b1[3] = {0.f, 0.f, 0.f};
                        dist_grad_block<KIND, D, GD, PRODUCT>(d2, &wq[G * 32 + half * 4], kDistTile, prm.clamp2, bP, b0, b1);
                        tP += bP;
                        t0 += b0;
#pragma unroll
                        for (int d = 0; d < D; ++d) t1[d] += b1[d];
                    }
                };
                if (tile_near) loop(std::true_type{});
                else loop(std::false_type{});
                accP += tP;
                S0 += t0;
#pragma unroll
                for (int d = 0; d < D; ++d) S1[d] += t1[d];
            }
        }

        if (wave_active) {
            accP += __shfl_xor(accP, 32, 64);          // the halves hold the two 16-column halves of every block
            S0 += __shfl_xor(S0, 32, 64);
#pragma unroll
            for (int d = 0; d < D; ++d) S1[d] += __shfl_xor(S1[d], 32, 64);
            const int i = wave_row0 + l31;
            if (half == 0 && i < row_end) {
                float dir[D];
#pragma unroll
                for (int d = 0; d < D; ++d) dir[d] = xs3[d] * S0 - S1[d];      // sum_j w_ij (xs_i - ys_j)[d]
                if (ns == 1) {
                    if (PRODUCT) prm.out[i] = accP;
                    const float gi = PRODUCT ? gp.gscale : gp.g[i] * gp.gscale;
#pragma unroll
                    for (int d = 0; d < D; ++d) gp.gx[(long)i * D + d] = gi * dir[d];
                } else {
                    float* part = sp.workspace + split * sp.split_stride + (long)i * kPart;
#pragma unroll
                    for (int d = 0; d < D; ++d) part[d] = dir[d];
                    if (PRODUCT) part[D] = accP;
                }
            }
        }
    }
}

}  // namespace glhip
