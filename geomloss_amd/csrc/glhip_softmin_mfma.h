// glhip_softmin_mfma.h — soft-min forward (p = 2, D <= 3) with the exponents formed on the matrix cores.
//
// The exponent of a pair in the expanded form is a length-4 dot product
//     u_ij - r_i = [a_i0, a_i1, a_i2, 1] . [yt_j0, yt_j1, yt_j2, H_j]
// i.e. a K = 4 fp32 GEMM tile, exactly what v_mfma_f32_16x16x4_f32 computes (fp32 in, fp32 accumulate,
// bitwise an fmaf chain).  On CDNA4 the MFMA pipe runs beside the VALU pipe, so moving the 3 FMAs and the
// "subtract the running max" (it becomes the C operand of the MFMA) off the VALU leaves it with only
// exp2 / add / max per pair: 6.6 -> ~2.6 VALU instructions per pair.
//
// Layout (one wavefront = RT row tiles of 16 rows; 4 wavefronts per workgroup = 64*RT rows):
//   A operand, per row tile: lane l holds A[i = l%16][k = l/16] = (a_i0 | a_i1 | a_i2 | 1)[k].
//   B operand: lane l holds B[k = l/16][j = l%16].  The LDS tile stores, per super-group of 64 columns,
//     one float4 per lane whose component g is the operand of column group g (16 columns); one
//     conflict-free ds_read_b128 feeds 4 MFMAs per row tile.
//   D: lane l, register r  <->  row 4*(l/16) + r, column l%16 of the group.  So each lane keeps the running
//     (max, sum) of 4 rows over its own 1/16th of the columns; nothing crosses lanes until the final
//     16-lane butterfly merge.
// The running max is lazy.  The first 64 columns initialise m_i exactly (max over the 64 columns, shared by
// the 16 lanes of the row).  After that a whole LDS tile (1024 columns) is accumulated speculatively with
// C = -m and NO per-pair max / compare at all; only at the end of the tile the tile sums are checked: if
// one passed kSumThr = 2^100 (a term ~2^80 above m arrived, or overflowed to +inf / NaN) the tile is redone
// from the untouched old sums with exact per-group maxima (C = 0).  m <= true max always, nothing can
// overflow unnoticed, and the hot loop is branch-free so the compiler can pipeline across column groups.
#pragma once

#include "glhip_mapreduce.h"
#include "glhip_softmin_ops.h"

namespace glhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaRT = 4;                            // 16-row tiles per wavefront (weighted-sum kernels)
constexpr int kMfmaRowsPerWave = kMfmaRT * 16;        // 64
constexpr int kMfmaRowsPerBlock = 4 * kMfmaRowsPerWave;   // 256
constexpr float kSumThr = 1.2676506e30f;            // 2^100: refresh the lazy max when a row sum passes it
constexpr float kMinusHuge = -3.0e38f;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 exp2v(f32x4 v) {
    return f32x4{fast_exp2(v.x), fast_exp2(v.y), fast_exp2(v.z), fast_exp2(v.w)};
}
__device__ __forceinline__ f32x4 maxv(f32x4 a, f32x4 b) {
    return f32x4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)};
}

template <int D, typename T, bool SPARSE, int RT = kMfmaRT>
__global__ void __launch_bounds__(kBlock)
softmin_fwd_mfma_kernel(SoftminParams<T> prm, Ranges rg, int N, int M, SplitInfo sp) {
    constexpr int kRowsPerWave = RT * 16;
    constexpr int kRowsPerBlock = 4 * kRowsPerWave;
    __shared__ f32x4 tileB[kTile];   // [super-group][lane] -> 4 column groups

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.y;
    const int split = blockIdx.z;
    const int ns = sp.n_splits;
    const int lk = lane >> 4;   // k index of this lane's A/B element; also the 4-row group of its D rows
    const int lj = lane & 15;

    int row_begin, row_end, q_begin, q_end;
    block_extent<SPARSE>(rg, N, kRowsPerBlock, row_begin, row_end, q_begin, q_end);

    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerBlock) {
        float centre[D];
        load_point<D, T>(prm.x, (long)b * N + row0, centre);

        // A operands of this wavefront's row tiles
        const int wave_row0 = row0 + wave * kRowsPerWave;
        float A[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int i = min(wave_row0 + rt * 16 + lj, row_end - 1);
            float v = (lk == 3) ? 1.0f : 0.0f;
            if (lk < D) v = (to_f32<T>(prm.x[((long)b * N + i) * D + lk]) - centre[lk < D ? lk : 0]) * prm.s2;
            A[rt] = v;
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
            for (int j0 = js; j0 < je; j0 += kTile) {
                const int n = min(kTile, je - j0);
                const int npad = (n + 63) & ~63;
                __syncthreads();
                // stage: column t -> component g of lanes (k*16 + j) of super-group G
                for (int t = tid; t < npad; t += kBlock) {
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
                    const int G = t >> 6, g = (t >> 4) & 3, j = t & 15;
                    float* base = reinterpret_cast<float*>(&tileB[G * 64]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) base[(k * 16 + j) * 4 + g] = rec[k];
                }
                __syncthreads();
                if (!wave_active) continue;

                const int nG = npad / 64;
                int G0 = 0;
                if (first_group) {
                    // exact initialisation on the first 64 columns; the max is shared by the 16 lanes of a row
                    const f32x4 B4 = tileB[lane];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x4 u0 = mfma4(A[rt], B4.x, zero), u1 = mfma4(A[rt], B4.y, zero);
                        const f32x4 u2 = mfma4(A[rt], B4.z, zero), u3 = mfma4(A[rt], B4.w, zero);
                        f32x4 um = maxv(maxv(u0, u1), maxv(u2, u3));
#pragma unroll
                        for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) um[r] = fmaxf(um[r], __shfl_xor(um[r], off, 64));
                        }
                        um = maxv(um, f32x4{kMinusHuge, kMinusHuge, kMinusHuge, kMinusHuge});   // -inf columns only
                        negm[rt] = -um;
                        ssum[rt] = (exp2v(u0 - um) + exp2v(u1 - um)) + (exp2v(u2 - um) + exp2v(u3 - um));
                    }
                    first_group = false;
                    G0 = 1;
                }

                // speculative, branch-free pass over the tile
                f32x4 stmp[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) stmp[rt] = zero;
                for (int G = G0; G < nG; ++G) {
                    const f32x4 B4 = tileB[G * 64 + lane];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x4 d0 = mfma4(A[rt], B4.x, negm[rt]);
                        const f32x4 d1 = mfma4(A[rt], B4.y, negm[rt]);
                        const f32x4 d2 = mfma4(A[rt], B4.z, negm[rt]);
                        const f32x4 d3 = mfma4(A[rt], B4.w, negm[rt]);
                        stmp[rt] += (exp2v(d0) + exp2v(d1)) + (exp2v(d2) + exp2v(d3));
                    }
                }
                float smax = 0.f;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    smax = fmaxf(fmaxf(smax, stmp[rt].x), fmaxf(fmaxf(stmp[rt].y, stmp[rt].z), stmp[rt].w));
                if (__any(!(smax < kSumThr))) {
                    // redo the tile from the old sums with exact per-group maxima
                    for (int G = G0; G < nG; ++G) {
                        const f32x4 B4 = tileB[G * 64 + lane];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const f32x4 u0 = mfma4(A[rt], B4.x, zero), u1 = mfma4(A[rt], B4.y, zero);
                            const f32x4 u2 = mfma4(A[rt], B4.z, zero), u3 = mfma4(A[rt], B4.w, zero);
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
            // merge the 16 column-lanes of every row, then lane (l%16 == r) finishes row 4*(l/16) + r
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
