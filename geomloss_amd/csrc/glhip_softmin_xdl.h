// glhip_softmin_xdl.h — soft-min forward (p = 2, D <= 3) with the exponents formed on the bf16 (XDL)
// matrix pipe at fp32 accuracy ("bf16 x 3" splitting).
//
// v_mfma_f32_16x16x4_f32 costs 34 cycles per 256 exponents — as much as the exp2 that follows.  The bf16
// instruction v_mfma_f32_16x16x32_bf16 costs ~17 for the same 256 outputs and has K = 32 slots.  Every fp32
// number is the exact sum of three bf16 numbers (8 + 8 + 8 significand bits, obtained by truncation), so
//     a * y = (a1 + a2 + a3)(y1 + y2 + y3) = a1y1 + a1y2 + a2y1 + a1y3 + a3y1 + a2y2 + a2y3 + a3y2   (+ a3y3 ~ 2^-32)
// — 8 exact bf16 products per coordinate, accumulated in fp32 by the MFMA.  K layout (4 blocks of 8 slots):
//   block d < D : A = [a1,a1,a2,a1,a3,a2,a2,a3] of a_id,  B = [y1,y2,y1,y3,y1,y2,y3,y2] of yt_jd
//   block 3     : A = [1,1,1,0,0,0,0,0],                  B = [H1,H2,H3,0,0,0,0,0]      (H_j split the same way)
// and C = -running max as in the fp32 kernel.  Lane l holds K-block l/16 of row / column l%16, so one
// 16-byte LDS record per (column, K-block) is the B operand as is.  D layout, lazy-max logic, column splits
// and the merge are those of glhip_softmin_mfma.h.
#pragma once

#include "glhip_softmin_mfma.h"

namespace glhip {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kTileX = 512;                 // columns per LDS tile: 32 groups x 64 lanes x 16 B = 32 KiB

union Pack16 { uint4 u; bf16x8 v; };

// three bf16 numbers (as the high halves of fp32 bit patterns) whose sum is v, by truncation
__device__ __forceinline__ void split3(float v, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    const uint32_t b1 = __float_as_uint(v) & 0xFFFF0000u;
    const float f1 = __uint_as_float(b1);
    float r = v - f1;                                  // exact
    if ((b1 & 0x7F800000u) == 0x7F800000u) r = 0.f;    // inf / nan stay in the first piece only
    const uint32_t b2 = __float_as_uint(r) & 0xFFFF0000u;
    const float r2 = r - __uint_as_float(b2);          // exact
    p1 = b1 >> 16;
    p2 = b2 >> 16;
    p3 = __float_as_uint(r2) >> 16;
}

__device__ __forceinline__ uint4 pack_a(float a) {     // [a1,a1,a2,a1,a3,a2,a2,a3]
    uint32_t p1, p2, p3;
    split3(a, p1, p2, p3);
    return uint4{p1 | (p1 << 16), p2 | (p1 << 16), p3 | (p2 << 16), p2 | (p3 << 16)};
}
__device__ __forceinline__ uint4 pack_y(float y) {     // [y1,y2,y1,y3,y1,y2,y3,y2]
    uint32_t p1, p2, p3;
    split3(y, p1, p2, p3);
    return uint4{p1 | (p2 << 16), p1 | (p3 << 16), p1 | (p2 << 16), p3 | (p2 << 16)};
}
__device__ __forceinline__ uint4 pack_h(float h) {     // [H1,H2,H3,0,0,0,0,0]
    uint32_t p1, p2, p3;
    split3(h, p1, p2, p3);
    return uint4{p1 | (p2 << 16), p3, 0u, 0u};
}

__device__ __forceinline__ f32x4 mfma_x(const uint4& a, const uint4& b, f32x4 c) {
    Pack16 pa, pb;
    pa.u = a;
    pb.u = b;
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa.v, pb.v, c, 0, 0, 0);
}

template <int D, typename T, bool SPARSE, int RT, int NW = 4>
__global__ void __launch_bounds__(NW * 64)
softmin_fwd_xdl_kernel(SoftminParams<T> prm, Ranges rg, int N, int M, SplitInfo sp) {
    constexpr int kRowsPerWave = RT * 16;
    constexpr int kRowsPerBlock = NW * kRowsPerWave;
    constexpr int kThreads = NW * 64;
    __shared__ uint4 tileX[(kTileX / 16) * 64];   // [column group][lane = kblock*16 + j]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    int bx, b, split;
    workgroup_coords(sp, bx, b, split);
    const int ns = sp.n_splits;
    const int lk = lane >> 4;
    const int lj = lane & 15;

    int row_begin, row_end, q_begin, q_end;
    block_extent<SPARSE>(rg, N, kRowsPerBlock, row_begin, row_end, q_begin, q_end, bx);

    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerBlock) {
        float centre[D];
        load_point<D, T>(prm.x, (long)b * N + row0, centre);

        const int wave_row0 = row0 + wave * kRowsPerWave;
        uint4 A[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int i = min(wave_row0 + rt * 16 + lj, row_end - 1);
            uint4 a = uint4{0u, 0u, 0u, 0u};
            if (lk < D) a = pack_a((to_f32<T>(prm.x[((long)b * N + i) * D + (lk < D ? lk : 0)]) - centre[lk < D ? lk : 0]) * prm.s2);
            else if (lk == 3) a = uint4{0x3F803F80u, 0x00003F80u, 0u, 0u};   // 1, 1, 1
            A[rt] = a;
        }
        f32x4 negm[RT], ssum[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            negm[rt] = f32x4{-kMinusHuge, -kMinusHuge, -kMinusHuge, -kMinusHuge};
            ssum[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const bool wave_active = wave_row0 < row_end;
        bool first_group = true;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

        for (int q = q_begin + (SPARSE ? split : 0); q < q_end; q += (SPARSE ? ns : 1)) {
            int js, je;
            column_interval<SPARSE>(rg, M, q, split, ns, js, je);
            for (int j0 = js; j0 < je; j0 += kTileX) {
                const int n = min(kTileX, je - j0);
                const int npad = (n + 63) & ~63;
                __syncthreads();
                for (int t = tid; t < npad; t += kThreads) {
                    float rec[4] = {0.f, 0.f, 0.f, kNegBig};
                    if (t < n) {
                        float yj[D];
                        load_point<D, T>(prm.y, (long)b * M + j0 + t, yj);
                        float n2 = 0.f;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            rec[d] = yj[d] - centre[d];
                            n2 = __builtin_fmaf(rec[d], rec[d], n2);
                        }
                        rec[3] = __builtin_fmaf(-0.5f * prm.s2, n2, dual_entry(prm, (long)b * M + j0 + t) * kLog2e);
                    }
                    uint4* base = &tileX[(t >> 4) * 64 + (t & 15)];
#pragma unroll
                    for (int d = 0; d < 3; ++d) base[d * 16] = (d < D) ? pack_y(rec[d]) : uint4{0u, 0u, 0u, 0u};
                    base[48] = pack_h(rec[3]);
                }
                __syncthreads();
                if (!wave_active) continue;

                const int nG = npad / 64;   // super-groups of 4 column groups
                int G0 = 0;
                if (first_group) {
                    uint4 Bq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) Bq[g] = tileX[g * 64 + lane];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x4 u0 = mfma_x(A[rt], Bq[0], zero), u1 = mfma_x(A[rt], Bq[1], zero);
                        const f32x4 u2 = mfma_x(A[rt], Bq[2], zero), u3 = mfma_x(A[rt], Bq[3], zero);
                        f32x4 um = maxv(maxv(u0, u1), maxv(u2, u3));
#pragma unroll
                        for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) um[r] = fmaxf(um[r], __shfl_xor(um[r], off, 64));
                        }
                        um = maxv(um, f32x4{kMinusHuge, kMinusHuge, kMinusHuge, kMinusHuge});
                        negm[rt] = -um;
                        ssum[rt] = (exp2v(u0 - um) + exp2v(u1 - um)) + (exp2v(u2 - um) + exp2v(u3 - um));
                    }
                    first_group = false;
                    G0 = 1;
                }

                f32x4 stmp[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) stmp[rt] = zero;
                for (int G = G0; G < nG; ++G) {
                    uint4 Bq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) Bq[g] = tileX[(G * 4 + g) * 64 + lane];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x4 d0 = mfma_x(A[rt], Bq[0], negm[rt]);
                        const f32x4 d1 = mfma_x(A[rt], Bq[1], negm[rt]);
                        const f32x4 d2 = mfma_x(A[rt], Bq[2], negm[rt]);
                        const f32x4 d3 = mfma_x(A[rt], Bq[3], negm[rt]);
                        stmp[rt] += (exp2v(d0) + exp2v(d1)) + (exp2v(d2) + exp2v(d3));
                    }
                }
                float smax = 0.f;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    smax = fmaxf(fmaxf(smax, stmp[rt].x), fmaxf(fmaxf(stmp[rt].y, stmp[rt].z), stmp[rt].w));
                if (__any(!(smax < kSumThr))) {
                    for (int G = G0; G < nG; ++G) {
                        uint4 Bq[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) Bq[g] = tileX[(G * 4 + g) * 64 + lane];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const f32x4 u0 = mfma_x(A[rt], Bq[0], zero), u1 = mfma_x(A[rt], Bq[1], zero);
                            const f32x4 u2 = mfma_x(A[rt], Bq[2], zero), u3 = mfma_x(A[rt], Bq[3], zero);
                            const f32x4 mold = -negm[rt];
                            const f32x4 mnew = maxv(mold, maxv(maxv(u0, u1), maxv(u2, u3)));
                            negm[rt] = -mnew;
                            ssum[rt] = ssum[rt] * exp2v(mold - mnew) +
                                       ((exp2v(u0 - mnew) + exp2v(u1 - mnew)) + (exp2v(u2 - mnew) + exp2v(u3 - mnew)));
                        }
                    }
                } else {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) ssum[rt] += stmp[rt];
                }
            }
        }

        if (wave_active) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 m = -negm[rt], s = ssum[rt];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    f32x4 m2, s2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        m2[r] = __shfl_xor(m[r], off, 64);
                        s2[r] = __shfl_xor(s[r], off, 64);
                    }
                    const f32x4 mn = maxv(m, m2);
                    s = s * exp2v(m - mn) + s2 * exp2v(m2 - mn);
                    m = mn;
                }
                // every lane of a 16-lane group now holds the merged (m, s) of the group's 4 rows; its first
                // lane writes them (no per-lane selects: keeps the epilogue free of divergent select chains)
                if (lj == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = wave_row0 + rt * 16 + lk * 4 + r;
                        if (i < row_end) {
                            float xi[D];
                            load_point<D, T>(prm.x, (long)b * N + i, xi);
                            float n2 = 0.f;
#pragma unroll
                            for (int d = 0; d < D; ++d) {
                                const float xt = xi[d] - centre[d];
                                n2 = __builtin_fmaf(xt, xt, n2);
                            }
                            const float mtot = __builtin_fmaf(-0.5f * prm.s2, n2, m[r]);   // r_i + m
                            if (ns == 1) {
                                prm.out[(long)b * N + i] = finish_value(prm, (long)b * N + i, mtot + fast_log2(s[r]));
                            } else {
                                float* dst = sp.workspace + split * sp.split_stride + ((long)b * N + i) * 2;
                                dst[0] = mtot;
                                dst[1] = s[r];
                            }
                        }
                    }
                }
            }
        }
    }
}

}  // namespace glhip
