// glhip_dist_x32.h — the reductions whose cost is a DISTANCE (not a squared distance): soft-min with p = 1, laplacian and
// energy kernel products — with the squared distance formed on the matrix cores (D <= 3).
//
// The VALU operators (SoftminFwdOp<.,1,true>, ConvOp<LAPLACIAN|ENERGY>) evaluate |x - y|^2 on explicit differences: 3 sub +
// 3 fma per pair before the square root — 13.3 / 10.1 / 9.1 VALU instructions per pair in all (profiles/r02_kernels_pmc.txt).
// Here the scaled squared distance of a 32 x 32 block of pairs comes out of the same chained pair of v_mfma_f32_32x32x16_bf16 as
// the exponent of the p = 2 soft-min (glhip_softmin_x32.h; every fp32 operand = 3 exact bf16 pieces):
//     d2_ij = |xs_i|^2 + |ys_j|^2 - 2 xs_i . ys_j      xs = t (x - c), ys = t (y - c)
//       K blocks d < D : y side [y1,y2,y1,y3,y1,y2,y3,y2]       x side pieces of -2 xs_d
//       K block 3      : y side [N1,N2,N3,1,1,1,0,0] (|ys|^2)    x side [1,1,1,n1,n2,n3,0,0] (|xs|^2)
// and, for the soft-min, a third MFMA broadcasts the per-column scalar to the lanes that hold the column's 16 rows:
//       K block 4      : y side [S1,S2,S3,1,1,1,0,0]             x side [1,1,1,m1,m2,m3,0,0]   (S = H_j, m = -running max)
// (the kernel products read their weight v_j from LDS instead — block_sum_lds — which frees 16 result registers: 8 waves per SIMD)
// What is left for the VALU per pair: max (the clamp of utils.py:61), v_sqrt_f32, and
//     soft-min p = 1 : sub, v_exp_f32, add          laplacian : v_exp_f32 (negated input), fma          energy : fma
// i.e. 5 / 4 / 3 instructions instead of 13 / 10 / 9.
//
// ACCURACY.  d2 carries the absolute error of the expanded form, ~2^-23 t^2 R^2 with R the distance of the two points from the
// centre c, so d = sqrt(d2) is off by ~2^-24 t R^2 / d: harmless for far pairs (R ~ d) but not for near pairs seen from a far
// centre.  Two measures keep that in check: (1) the kernel is only used on BLOCK-SPARSE launches whose row blocks are spatially
// compact (the voxel clusters of the multiscale backends, or the voxel sort the Python drivers apply to large dense launches),
// with c = the mean of the rows of the workgroup's pass, so that near pairs have R <~ half a cluster diameter; (2) pairs closer
// than R / 16 (d2 < 2^-8 |xs_i|^2: ~0.4 % of the blocks) are re-evaluated on explicit differences (exact_near_pairs).  What is
// left is an error <= ~2^-18 R on the distance of a pair, i.e. < 1e-6 on a potential for clusters of diameter 0.2.
// The caller opts in with GLHIP_FLAG_MFMA_DIST; everything else stays on the direct-difference operators.
#pragma once

#include <type_traits>

#include "glhip_kconv_ops.h"
#include "glhip_softmin_x32.h"

namespace glhip {

enum DistMode { DM_SOFTMIN_P1 = 0, DM_LAPLACIAN = 1, DM_ENERGY = 2 };

template <typename T>
struct DistParams {
    const T* x;          // (N,D)
    const T* y;          // (M,D)
    const float* s;      // (M): dual vector h (soft-min) or weights v (kernel products)
    const float* pot;    // soft-min, fused half-step: h_j := s_j + pot_scale * pot_j (or NULL)
    const float* prev;   // soft-min, fused half-step: out_i := alpha * f_i + beta * prev_i (or NULL)
    float* out;          // (N)
    float t;             // coordinate scale: log2(e)/eps (p = 1), log2(e)/blur (laplacian), 1 (energy)
    float clamp2;        // 1e-8 t^2
    float out_scale;     // soft-min: -eps ln 2
    float pot_scale, alpha, beta;
    float guard;         // pairs with d2 < guard * |xs_i|^2 are re-evaluated on explicit differences (2^-8 by default)
};

constexpr int kDistRec = 5;                 // 16-byte records per column
constexpr int kDistTile = 512;              // columns per LDS tile: 512 x 5 x 16 B = 40 KiB

// a scaled column coordinate back from its [y1,y2,y1,y3,y1,y2,y3,y2] record (the three pieces sum to the fp32 value exactly)
__device__ __forceinline__ float unpack_y(const uint4& r) {
    return (__uint_as_float(r.x << 16) + __uint_as_float(r.x & 0xFFFF0000u)) + __uint_as_float(r.y & 0xFFFF0000u);
}

// Near pairs.  The MFMA squared distance of a pair much closer than its offset from the centre (d < R / 16) has lost too many
// bits to the cancellation |xs|^2 + |ys|^2 - 2 xs.ys; such pairs are rare (1024 pi/3 (1/16)^3 (R/L)^3 per block for clouds of
// size L: ~0.4 % of the blocks at R/L = 0.1) and are recomputed here on explicit differences of the coordinates kept in LDS.
// Register k of lane l holds column (k/4)*8 + (l/32)*4 + k%4 of the group (32x32 MFMA result layout).
// The exact value is floored at clamp2 here (utils.py:61), so that the main loop can take square roots without a clamp: every
// squared distance it sees is either >= thr >= clamp2 as computed, or exact and floored.
template <int D>
__device__ __forceinline__ void exact_near_pairs(f32x16& d2, float thr, float clamp2, const uint4* __restrict__ group, const float (&xs)[3],
                                                 int half) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (d2[k] < thr) {
            const int col = (k >> 2) * 8 + half * 4 + (k & 3);
            float e = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float df = xs[d] - unpack_y(group[d * 32 + col]);
                e = __builtin_fmaf(df, df, e);
            }
            d2[k] = fmaxf(e, clamp2);
        }
    }
}

__device__ __forceinline__ float min16(const f32x16& v) {
    const float a = fminf(fminf(v[0], v[1]), v[2]), b = fminf(fminf(v[3], v[4]), v[5]), c = fminf(fminf(v[6], v[7]), v[8]);
    const float d = fminf(fminf(v[9], v[10]), v[11]), e = fminf(fminf(v[12], v[13]), v[14]);
    return fminf(fminf(fminf(a, b), fminf(c, d)), fminf(e, v[15]));
}

// reduction of one 32 x 32 block: d2 (scaled squared distances), sb (per-column scalar, minus the running max for the soft-min)
// CLAMP = false: the caller guarantees d2 >= clamp2 (see the near-pair logic of the kernel): |d2| only guards the square root
// against a stray negative rounding residue (the absolute value is a free source modifier)
// FAMILY: |.| = m rsq(m) instead of sqrt(m) — the expression of the product-and-gradient kernels (glhip_dist_grad_x32.h), for the
// other terms of a kernel norm whose gradient is on (GLHIP_FLAG_GRAD_FAMILY)
template <bool CLAMP, bool FAMILY = false>
__device__ __forceinline__ float dist_of(float d2, float clamp2) {
    // v_med3_f32: the clamp in ONE instruction (fmaxf costs two: the compiler canonicalises the MFMA result first)
    const float m = CLAMP ? __builtin_amdgcn_fmed3f(d2, clamp2, 3.0e38f) : __builtin_fabsf(d2);
    return FAMILY ? m * fast_rsq(m) : fast_sqrt(m);
}

template <int MODE, bool CLAMP = true>
__device__ __forceinline__ float block_sum(const f32x16& d2, const f32x16& sb, float clamp2) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float dist = dist_of<CLAMP>(d2[k], clamp2);
        if (MODE == DM_SOFTMIN_P1) acc[k & 3] += fast_exp2(sb[k] - dist);
        else if (MODE == DM_LAPLACIAN) acc[k & 3] = __builtin_fmaf(fast_exp2(-dist), sb[k], acc[k & 3]);
        else acc[k & 3] = __builtin_fmaf(-dist, sb[k], acc[k & 3]);
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// kernel products: the per-column weight read straight from LDS (4 broadcast ds_read_b128 per block) instead of a third MFMA —
// 16 result registers and one operand less, which is what lets 8 waves per SIMD fit (<= 64 VGPRs; 4 records + 4 bytes per column
// = 34 KiB of LDS per workgroup).  sg = &weights[first column of the group + 4 * half]: register k <-> column (k/4)*8 + 4*half + k%4.
template <int MODE, bool CLAMP = true, bool FAMILY = false>
__device__ __forceinline__ float block_sum_lds(const f32x16& d2, const float* __restrict__ sg, float clamp2) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 s4 = *reinterpret_cast<const float4*>(sg + q * 8);
        const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dist = dist_of<CLAMP, FAMILY>(d2[q * 4 + r], clamp2);
            if (MODE == DM_LAPLACIAN) acc[r] = __builtin_fmaf(fast_exp2(-dist), sv[r], acc[r]);
            else acc[r] = __builtin_fmaf(-dist, sv[r], acc[r]);
        }
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

template <int MODE, int D, typename T, int NW, bool FAMILY = false>
__global__ void __launch_bounds__(NW * 64, MODE == DM_SOFTMIN_P1 ? 6 : 8)     // kernel products: 8 waves per SIMD (<= 64 VGPRs); soft-min: 6 (<= 80)
dist_x32_kernel(DistParams<T> prm, Ranges rg, int N, int M, SplitInfo sp) {
    static_assert(!FAMILY || MODE != DM_SOFTMIN_P1, "FAMILY is a variant of the kernel products");
    constexpr int kRowsPerBlock = NW * 32;
    constexpr int kThreads = NW * 64;
    constexpr bool kWeightsInLds = MODE != DM_SOFTMIN_P1;
    constexpr int REC = kWeightsInLds ? 4 : kDistRec;
    __shared__ uint4 tile[kDistTile * REC];           // [column group of 32][K block 0..REC-1][column]
    __shared__ float weights[kWeightsInLds ? kDistTile : 4];
    __shared__ float csum[NW][4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.z;
    const int ns = sp.n_splits;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int rec0 = half * 32 + l31;     // K block `half` of column l31 inside a group (+64: K block half + 2)

    int row_begin, row_end, q_begin, q_end;
    block_extent<true>(rg, N, kRowsPerBlock, row_begin, row_end, q_begin, q_end, blockIdx.x);

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint4 kOnes = uint4{0x3F803F80u, 0x00003F80u, 0u, 0u};   // [1,1,1,0,...]
    const uint4 kZero = uint4{0u, 0u, 0u, 0u};

    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerBlock) {
        // centre of the pass = mean of its rows (half the worst-case offset of "the first row"): the error of a near pair's
        // distance grows with the square of the offsets from the centre
        float centre[D];
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
            __syncthreads();      // previous pass done with csum
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
        uint4 Xlo, Xhi, Xs;
        float xs3[3] = {0.f, 0.f, 0.f}, thr;
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
            thr = prm.guard * n2;
            thr = fmaxf(thr, prm.clamp2);                  // everything below the floor of utils.py:61 takes the exact path too
            const uint4 p0 = pack_a(a[0]), p1 = (D > 1) ? pack_a(a[1]) : kZero, p2 = (D > 2) ? pack_a(a[2]) : kZero;
            Xlo = select_u4(half != 0, p1, p0);
            Xhi = select_u4(half != 0, pack_negmax(-n2), p2);   // block 3 carries + |xs|^2
            Xs = select_u4(half != 0, kZero, kOnes);            // block 4: the scalar itself (soft-min: rewritten with the running max)
        }
        // a column can only be a "near" partner (d < R_i / 16) of some row of this pass if it lies within 17/16 of the largest row
        // offset from the centre: tiles without such a column (nearly all of them once the columns are spatially sorted too) skip
        // the near-pair test altogether
        float near2;
        {
            float r2 = wave_active ? prm.guard * xs3[0] * xs3[0] + prm.guard * (xs3[1] * xs3[1] + xs3[2] * xs3[2]) : 0.f;   // guard |xs|^2
            for (int off = 32; off > 0; off >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, off, 64));
            __syncthreads();
            if (lane == 0) csum[wave][3] = r2;
            __syncthreads();
            float tot = 0.f;
            for (int w = 0; w < NW; ++w) tot = fmaxf(tot, csum[w][3]);
            // thr = max(guard |xs|^2, clamp2).  A column within 17/16 of the largest row offset — or within the clamp radius of it —
            // may form a pair below thr; every other tile is far enough for its squared distances to be >= 4 clamp2 as computed
            const float rmax = fast_sqrt(tot / fmaxf(prm.guard, 1e-30f)), reach = 1.07f * rmax + 2.f * fast_sqrt(prm.clamp2);
            near2 = (prm.guard > 0.f) ? reach * reach : -1.f;
        }
        float m = kMinusHuge, ssum = 0.f;                  // soft-min: lazy running max and sum;  products: ssum only
        bool first_group = true;

        for (int q = q_begin + split; q < q_end; q += ns) {
            const int js = rg.redranges_j[2 * q], je = rg.redranges_j[2 * q + 1];
            for (int j0 = js; j0 < je; j0 += kDistTile) {
                const int n = min(kDistTile, je - j0);
                const int npad = (n + 31) & ~31;
                __syncthreads();
                int near = 0;
                for (int t = tid; t < npad; t += kThreads) {
                    // padding columns: zero weight; FAMILY: also infinitely far (0 * rsq(0) is not a number)
                    float ys[3] = {0.f, 0.f, 0.f}, n2 = (FAMILY && t >= n) ? 1.0e30f : 0.f, sj = (MODE == DM_SOFTMIN_P1) ? kNegBig : 0.f;
                    if (t < n) {
                        float yj[D];
                        load_point<D, T>(prm.y, j0 + t, yj);
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            ys[d] = (yj[d] - centre[d]) * prm.t;
                            n2 = __builtin_fmaf(ys[d], ys[d], n2);
                        }
                        sj = prm.s[j0 + t];
                        if (MODE == DM_SOFTMIN_P1) {
                            if (prm.pot) sj = __builtin_fmaf(prm.pot[j0 + t], prm.pot_scale, sj);
                            sj *= kLog2e;
                        }
                    }
                    uint4* base = &tile[(t >> 5) * (32 * REC) + (t & 31)];
#pragma unroll
                    for (int d = 0; d < 3; ++d) base[d * 32] = (d < D) ? pack_y(ys[d]) : kZero;
                    base[3 * 32] = pack_h1(n2);
                    if constexpr (kWeightsInLds) weights[t] = sj;
                    else base[4 * 32] = pack_h1(sj);
                    near |= (t < n && n2 < near2) ? 1 : 0;
                }
                const bool tile_near = __syncthreads_or(near) != 0;      // workgroup-uniform
                if (!wave_active) continue;

                const int nG = npad / 32;
                int G0 = 0;
                if (MODE == DM_SOFTMIN_P1 && first_group) {      // exact maximum over the first 32 columns
                    const uint4* g = &tile[0];
                    f32x16 d2 = mfma_x32(g[rec0], Xlo, zero16);
                    d2 = mfma_x32(g[64 + rec0], Xhi, d2);
                    if (tile_near && __any(min16(d2) < thr)) exact_near_pairs<D>(d2, thr, prm.clamp2, g, xs3, half);
                    const f32x16 sb = mfma_x32(select_u4(half != 0, kZero, g[128 + l31]), Xs, zero16);
                    float um = kMinusHuge;
#pragma unroll
                    for (int k = 0; k < 16; ++k) um = fmaxf(um, sb[k] - fast_sqrt(__builtin_amdgcn_fmed3f(d2[k], prm.clamp2, 3.0e38f)));
                    um = fmaxf(um, __shfl_xor(um, 32, 64));
                    m = um;
                    Xs = select_u4(half == 0, pack_negmax(m), Xs);
                    ssum = block_sum<MODE>(d2, mfma_x32(select_u4(half != 0, kZero, g[128 + l31]), Xs, zero16), prm.clamp2);
                    first_group = false;
                    G0 = 1;
                }
                // guarded: the tile may hold pairs below thr (near pairs, or pairs inside the clamp radius): test every block.
                // clamped: only when the near-pair guard is switched off (GLHIP_DIST_GUARD=0), otherwise no squared distance that
                // reaches the square root is below clamp2 (see exact_near_pairs) and the clamp instruction is dropped.
                auto main_loop = [&](auto guarded, auto clamped) {
                    constexpr bool CL = decltype(clamped)::value;
                    float st = 0.f;
                    for (int G = G0; G < nG; ++G) {
                        const uint4* g = &tile[G * (32 * REC)];
                        f32x16 d2 = mfma_x32(g[rec0], Xlo, zero16);
                        d2 = mfma_x32(g[64 + rec0], Xhi, d2);
                        if constexpr (decltype(guarded)::value) {
                            if (__any(min16(d2) < thr)) exact_near_pairs<D>(d2, thr, prm.clamp2, g, xs3, half);
                        }
                        if constexpr (kWeightsInLds) {
                            st += block_sum_lds<MODE, CL, FAMILY>(d2, &weights[G * 32 + half * 4], prm.clamp2);
                        } else {
                            const f32x16 sb = mfma_x32(select_u4(half != 0, kZero, g[128 + l31]), Xs, zero16);
                            st += block_sum<MODE, CL>(d2, sb, prm.clamp2);
                        }
                    }
                    return st;
                };
                const float stmp = !(prm.guard > 0.f) ? main_loop(std::false_type{}, std::true_type{})
                                   : tile_near        ? main_loop(std::true_type{}, std::false_type{})
                                                      : main_loop(std::false_type{}, std::false_type{});
                if (MODE == DM_SOFTMIN_P1 && __any(!(stmp < kSumThr))) {
                    // a term far above the lazy max arrived (or inf / NaN): redo the tile with exact per-group maxima
                    const uint4 plain = select_u4(half != 0, kZero, kOnes);
                    for (int G = G0; G < nG; ++G) {
                        const uint4* g = &tile[G * (32 * REC)];
                        f32x16 d2 = mfma_x32(g[rec0], Xlo, zero16);
                        d2 = mfma_x32(g[64 + rec0], Xhi, d2);
                        if (tile_near && __any(min16(d2) < thr)) exact_near_pairs<D>(d2, thr, prm.clamp2, g, xs3, half);
                        const f32x16 sb = mfma_x32(select_u4(half != 0, kZero, g[128 + l31]), plain, zero16);
                        float u[16], um = kMinusHuge;
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            u[k] = sb[k] - fast_sqrt(__builtin_amdgcn_fmed3f(d2[k], prm.clamp2, 3.0e38f));
                            um = fmaxf(um, u[k]);
                        }
                        um = fmaxf(um, __shfl_xor(um, 32, 64));
                        const float mnew = fmaxf(m, um);
                        float s2 = 0.f;
#pragma unroll
                        for (int k = 0; k < 16; ++k) s2 += fast_exp2(u[k] - mnew);
                        ssum = ssum * fast_exp2(m - mnew) + s2;
                        m = mnew;
                    }
                    Xs = select_u4(half == 0, pack_negmax(m), Xs);
                } else {
                    ssum += stmp;
                }
            }
        }

        if (wave_active) {
            const float s = ssum + __shfl_xor(ssum, 32, 64);     // the two halves hold the two 16-column halves of every block
            const int i = wave_row0 + l31;
            if (half == 0 && i < row_end) {
                if (MODE == DM_SOFTMIN_P1) {
                    if (ns == 1) {
                        float f = prm.alpha * (prm.out_scale * (m + fast_log2(s)));
                        if (prm.prev) f = __builtin_fmaf(prm.beta, prm.prev[i], f);
                        prm.out[i] = f;
                    } else {
                        float* dst = sp.workspace + split * sp.split_stride + (long)i * 2;
                        dst[0] = m;
                        dst[1] = s;
                    }
                } else {
                    if (ns == 1) prm.out[i] = s;
                    else sp.workspace[split * sp.split_stride + i] = s;
                }
            }
        }
    }
}

}  // namespace glhip
