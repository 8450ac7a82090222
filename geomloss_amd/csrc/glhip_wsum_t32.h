// glhip_wsum_t32.h — the weighted-sum reductions (soft-min gradient, gaussian gradient, gaussian product + gradient) on TRANSPOSED
// 32 x 32 matrix-core blocks, for 1 <= D <= 16.
//
//     R_i[c] = sum_j  2^( [a_i,1].[yt_j,H_j] + C_i ) * q_j[c]          (modes and the meaning of C_i, q_j: WsumMode, glhip_wsum_mfma.h)
//
// glhip_wsum_mfma.h (16x16x32 MFMAs) and the rejected 32x32 variant of glhip_wsum_x32.h keep the MFMA rows = rows x_i: a lane
// then holds 4 (16) different ROWS, each with its own D + 1 accumulators — 64 accumulator registers at D = 3, out of reach beyond.
// Here the block is transposed exactly as in the forward kernels (glhip_softmin_x32.h / _xd.h): the MFMA "A" rows are 32 columns
// y_j from LDS, the "B" columns are 32 rows x_i in registers, so lane l owns ONE row (l % 32) and its 16 result registers are 16
// columns.  The accumulators are D + 1 (soft-min: D + mass) registers per row tile whatever D is, the per-row constant C_i rides
// in the spare K slots of the scalar block, and what the transposition costs is that the small per-column vectors q_j are no
// longer "one register per lane": they are read from LDS, component by component, as broadcast float4 (4 consecutive columns of
// the lane's half: register k <-> column (k / 4) * 8 + 4 * half + k % 4) — 4 (D + 1) ds_read_b128 per 32-column group, shared by the
// RT row tiles of the wavefront.  Per 1024 pairs and row tile: NM MFMAs, 16 v_exp_f32, 16 (D + 1) v_fma_f32 / v_add_f32.
//
// Used for 4 <= D <= 16 (before: the one-thread-per-row VALU kernel of glhip_generic.h).  For D <= 3 it was measured against the
// 16x16x32 kernel and is no faster (soft-min gradient at 1e6: 158 vs 148 ms; gaussian gradient 159 vs 163 ms;
// profiles/r03_grad_kernels_ab.txt): 126 VGPRs = 4 waves per SIMD, where the 16x16x32 kernel runs 5 — the D <= 3 gradients stay there.
#pragma once

#include "glhip_softmin_xd.h"
#include "glhip_wsum_mfma.h"

namespace glhip {

template <int D, int L = XL_BF16X3>
struct T32Shape {      // the K layouts of XdShape (glhip_softmin_xd.h): 6 slots per coordinate (bf16 x 3) or 3 (f16 x 2)
    static constexpr int NM = XdShape<D, L>::NM;
    static constexpr int NBP = 2 * NM;
    // columns per LDS tile: (16 NBP + 4 (D + 1)) bytes each, sized so that LDS never caps the kernel below its 4 waves per SIMD
    // (512 columns left the D = 5 .. 7 kernels two 4-wave workgroups per CU: soft-min gradient at D = 5 230 -> 214 ms)
    static constexpr int kTile = NBP <= 6 ? 256 : 128;
};

template <int MODE, int D>
struct T32Q {           // components of q_j kept in LDS, accumulators per row
    static constexpr int NQ = (MODE == WS_SOFTMIN_BWD) ? D : D + 1;     // soft-min: yt_j (the mass needs no q);  gaussian: (v yt_j, v)
    static constexpr int NA = D + 1;
};

template <int MODE, int D, typename T, bool SPARSE, int RT, int NW, int L = XL_BF16X3>
__global__ void __launch_bounds__(NW * 64, 4)      // <= 128 VGPRs: two 8-wave (four 4-wave) workgroups per CU; without the bound the
                                                    // record assembly of the row pass pushes D = 16 to 131 VGPRs = one workgroup per CU (2x slower)
wsum_t32_kernel(WsumParams<T> prm, Ranges rg, int N, int M, SplitInfo sp) {
    using S = T32Shape<D, L>;
    constexpr int NM = S::NM, NBP = S::NBP, kTileD = S::kTile;
    const float xscale = (L == XL_F16X2) ? __builtin_sqrtf(prm.s2) : prm.s2;      // f16 x 2: sqrt(s) on both sides (glhip_softmin_xd.h)
    constexpr int NQ = T32Q<MODE, D>::NQ, NA = T32Q<MODE, D>::NA;
    constexpr int kRowsPerWave = RT * 32;
    constexpr int kRowsPerBlock = NW * kRowsPerWave;
    constexpr int kThreads = NW * 64;
    static_assert(kRowsPerBlock == kMfmaRowsPerBlock, "the merge kernels tile rows (and centre their partials) in blocks of 256");
    static_assert(MODE == WS_SOFTMIN_BWD || MODE == WS_GAUSS_BWD || MODE == WS_GAUSS_FWDGRAD, "product-only mode: glhip_softmin_xd.h");
    __shared__ uint4 tile[kTileD * NBP];          // [column group of 32][K block][column]
    __shared__ __attribute__((aligned(16))) float tileQ[NQ * kTileD];          // [component][column]: read back as float4

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, b, split;
    workgroup_coords(sp, bx, b, split);
    const int ns = sp.n_splits;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int rec0 = half * 32 + l31;

    int row_begin, row_end, q_begin, q_end;
    block_extent<SPARSE>(rg, N, kRowsPerBlock, row_begin, row_end, q_begin, q_end, bx);

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerBlock) {
        float centre[D];
        load_point<D, T>(prm.x, (long)b * N + row0, centre);

        const int wave_row0 = row0 + wave * kRowsPerWave;
        const bool wave_active = wave_row0 < row_end;
        uint4 X[RT][NM];
        float acc[RT][NA];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int i = min(wave_row0 + rt * 32 + l31, row_end - 1);
            float xi[D];
            load_point<D, T>(prm.x, (long)b * N + i, xi);
            float a[D], n2 = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float xt = xi[d] - centre[d];
                n2 = __builtin_fmaf(xt, xt, n2);
                a[d] = xt * xscale;
            }
            // per-row constant of the exponent: r_i = -s/2 |xt_i|^2, minus (LSE2_i - r_i)'s other half for the soft-min gradient:
            //   C_i = r_i - LSE2_i, LSE2_i = fwd_i / out_scale (+ tscale: value-and-gradient mode, `fwd` is a guess and tscale the margin)
            float cst = -0.5f * prm.s2 * n2;
            if (MODE == WS_SOFTMIN_BWD) cst -= prm.fwd[(long)b * N + i] / prm.out_scale + prm.tscale;
            if (L == XL_F16X2) cst = __builtin_fminf(__builtin_fmaxf(cst, kH2Floor), -kH2Floor);
#pragma unroll
            for (int mm = 0; mm < NM; ++mm)     // scalar item [1,1,1,c1,c2,c3] with c = cst, then the slots of every coordinate (xd_record_of)
                X[rt][mm] = select_u4(half != 0, xd_record_of<D, true, L>(2 * mm + 1, cst, a), xd_record_of<D, true, L>(2 * mm, cst, a));
#pragma unroll
            for (int c = 0; c < NA; ++c) acc[rt][c] = 0.f;
        }

        for (int q = q_begin + (SPARSE ? split : 0); q < q_end; q += (SPARSE ? ns : 1)) {
            int js, je;
            column_interval<SPARSE>(rg, M, q, split, ns, js, je);
            for (int j0 = js; j0 < je; j0 += kTileD) {
                const int n = min(kTileD, je - j0);
                const int npad = (n + 31) & ~31;
                __syncthreads();
                for (int t = tid; t < npad; t += kThreads) {
                    float yt[D], H = kNegBig, sj = 0.f;
#pragma unroll
                    for (int d = 0; d < D; ++d) yt[d] = 0.f;
                    if (t < n) {
                        const long col = (long)b * M + j0 + t;
                        float yj[D];
                        load_point<D, T>(prm.y, col, yj);
                        float n2 = 0.f;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            yt[d] = yj[d] - centre[d];
                            n2 = __builtin_fmaf(yt[d], yt[d], n2);
                        }
                        sj = prm.s[col];
                        H = (MODE == WS_SOFTMIN_BWD) ? __builtin_fmaf(-0.5f * prm.s2, n2, sj * kLog2e) : -0.5f * prm.s2 * n2;
                    }
#pragma unroll
                    for (int d = 0; d < D; ++d) tileQ[d * kTileD + t] = (MODE == WS_SOFTMIN_BWD) ? yt[d] : sj * yt[d];
                    if (MODE != WS_SOFTMIN_BWD) tileQ[D * kTileD + t] = (t < n) ? sj : 0.f;
                    if (L == XL_F16X2) {
#pragma unroll
                        for (int d = 0; d < D; ++d) yt[d] *= xscale;
                        H = __builtin_fmaxf(H, kH2Floor);
                    }
                    uint4* base = &tile[(t >> 5) * (32 * NBP) + (t & 31)];
#pragma unroll
                    for (int r = 0; r < NBP; ++r) base[r * 32] = xd_record_of<D, false, L>(r, H, yt);     // record by record: few live pieces
                }
                __syncthreads();
                if (!wave_active) continue;

                for (int G = 0; G < npad / 32; ++G) {
                    const uint4* g = &tile[G * (32 * NBP)];
                    f32x16 w[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x16 u = xd_block<NM, NBP, L>(g, rec0, X[rt], zero16);
#pragma unroll
                        for (int k = 0; k < 16; ++k) w[rt][k] = fast_exp2(u[k]);
                    }
                    const float* qg = &tileQ[G * 32 + half * 4];
#pragma unroll
                    for (int c = 0; c < NQ; ++c) {
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const float4 q4 = *reinterpret_cast<const float4*>(qg + c * kTileD + qq * 8);
                            const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[rt][c] = __builtin_fmaf(w[rt][qq * 4 + r], qv[r], acc[rt][c]);
                            }
                        }
                    }
                    if (MODE == WS_SOFTMIN_BWD) {      // the plan mass: sum of the weights themselves
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int k = 0; k < 16; ++k) s4[k & 3] += w[rt][k];
                            acc[rt][D] += (s4[0] + s4[1]) + (s4[2] + s4[3]);
                        }
                    }
                }
            }
        }

        if (wave_active) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float a_[NA];
#pragma unroll
                for (int c = 0; c < NA; ++c) a_[c] = acc[rt][c] + __shfl_xor(acc[rt][c], 32, 64);   // the two 16-column halves
                const int i = wave_row0 + rt * 32 + l31;
                if (half == 0 && i < row_end) {
                    const long idx = (long)b * N + i;
                    float xt[D];
                    {
                        float xi[D];
                        load_point<D, T>(prm.x, idx, xi);
#pragma unroll
                        for (int d = 0; d < D; ++d) xt[d] = xi[d] - centre[d];
                    }
                    float* part = sp.workspace + split * sp.split_stride + idx * (D + 1 - (MODE == WS_GAUSS_BWD ? 1 : 0));
                    if (MODE == WS_SOFTMIN_BWD) {
                        // a_[d] = sum_j P_ij yt_jd, a_[D] = sum_j P_ij;  d f_i / d x_i = x_i - sum_j P_ij y_j / sum_j P_ij
                        if (ns == 1) {
                            const float gi = prm.g ? prm.g[idx] : 1.f;
                            const float inv = (a_[D] > 0.f) ? 1.0f / a_[D] : 0.f;
#pragma unroll
                            for (int d = 0; d < D; ++d) prm.gx[idx * D + d] = gi * (xt[d] - a_[d] * inv);
                            if (prm.out)   // value-and-gradient mode: the mass turns the guess into the exact soft-min
                                prm.out[idx] = prm.fwd[idx] + prm.out_scale * (prm.tscale + fast_log2(a_[D]));
                        } else {
#pragma unroll
                            for (int c = 0; c < NA; ++c) part[c] = a_[c];      // SoftminBwdOp::merge_row: relative to x[row0], as here
                        }
                    } else if (MODE == WS_GAUSS_FWDGRAD) {
                        // a_[d] = sum_j v_j k_ij yt_jd, a_[D] = sum_j v_j k_ij = the product;  d out_i / d x_i = gscale (xt S0 - S1)
                        if (ns == 1) {
                            prm.out[idx] = a_[D];
#pragma unroll
                            for (int d = 0; d < D; ++d) prm.gx[idx * D + d] = prm.gscale * (xt[d] * a_[D] - a_[d]);
                        } else {
#pragma unroll
                            for (int d = 0; d < D; ++d) part[d] = prm.tscale * (xt[d] * a_[D] - a_[d]);
                            part[D] = a_[D];
                        }
                    } else {
                        if (ns == 1) {
                            const float gi = prm.g[idx] * prm.gscale;
#pragma unroll
                            for (int d = 0; d < D; ++d) prm.gx[idx * D + d] = gi * (xt[d] * a_[D] - a_[d]);
                        } else {
#pragma unroll
                            for (int d = 0; d < D; ++d) part[d] = prm.tscale * (xt[d] * a_[D] - a_[d]);
                        }
                    }
                }
            }
        }
    }
}


// ---- the weighted sums themselves on the matrix cores (round 5; soft-min gradient, GLHIP_FLAG_F16X2, D >= kWqMinD) --------------------
//
// In the kernel above the exponents come from NM MFMAs per 1024 pairs, but the sums R_i[c] = sum_j w_ij q_j[c] are 16 (D + 1) v_fma_f32
// per 1024 pairs and row tile: 544 VALU cycles at D = 16 next to 128 for the exponentials — the gradient kernels of 4 <= D <= 16 were
// VALU-bound on a contraction (round-4 review: 0.43 of nominal issue at D = 16).  Here the contraction is a second set of MFMAs:
//     out[c][i] = sum_j Q[c][j] W[j][i],    A = Q (component c = MFMA row, from LDS),  B = W (row i = MFMA column, IN REGISTERS)
// and it needs no data movement at all, because the exponent block is already transposed: lane (half h, i) holds the weights of row i
// for the 16 columns col(h, r) = 8 (r / 4) + 4 h + r % 4 — and the B operand of v_mfma_f32_32x32x16_f16 wants, from lane (h, n = i),
// eight K values: registers r = 0..7 are the K slots 8 h .. 8 h + 7 of a first instruction, r = 8..15 of a second one.  Which column
// a K slot means is ours to define, as long as the A operand (staged by us) uses the same order.  Precision: w (scaled by 2^13 so
// that its low piece and the small weights stay clear of the f16 subnormals) and q are split in two f16 pieces each, three
// products (hi hi, hi lo, lo hi): six MFMAs per 32 x 32 block whatever D is, with a fresh accumulator — the MFMA truncates its
// final rounding (tools/ubench/mfma_round.hip), a bias that 31 250 chained blocks would turn into 1e-3 — and one v_add_f32 per
// useful component register folds the block into the row's running sums (round to nearest).  Per 1024 pairs and row tile:
// NM + 6 MFMAs, 16 v_exp_f32, ~48 conversion instructions, <= 16 v_add_f32 — independent of D.
//   components: c < D: sqrt(s) yt_j[c] (the same sqrt(s) as the exponent operands: in f16 range by the flag's contract), c = D: 1
//   (the plan mass), c > D: 0.  Result register r of lane (h, i) <-> component c = (r & 3) + 8 (r >> 2) + 4 h.
constexpr int kWqShift = 13;      // w' = 2^13 w <= 8192 for w <= 1 (a plan weight, or a weight relative to a bound of the answer)

template <int D>
struct T32QShape {
    static constexpr int NM = XdShape<D, XL_F16X2>::NM;
    static constexpr int NBP = 2 * NM;
    static constexpr int kTile = 128;                    // columns per LDS tile: 4 groups
    static constexpr int kQRecs = 4 * 64;                // 16-byte records of Q per column group: [piece hi / lo][instruction 0 / 1][lane]
};

template <int D, typename T, bool SPARSE, int NW>
__global__ void __launch_bounds__(NW * 64, 4)
wsum_t32q_kernel(WsumParams<T> prm, Ranges rg, int N, int M, SplitInfo sp) {
    using S = T32QShape<D>;
    constexpr int L = XL_F16X2;
    constexpr int NM = S::NM, NBP = S::NBP, kTileD = S::kTile, kGroups = kTileD / 32;
    constexpr int kRowsPerBlock = NW * 32;
    constexpr int kThreads = NW * 64;
    static_assert(kRowsPerBlock == kMfmaRowsPerBlock, "the merge kernels tile rows (and centre their partials) in blocks of 256");
    static_assert(D >= 1 && D <= 16, "components 0..D must fit the 32 MFMA rows (and the scalar layout D <= 16)");
    __shared__ uint4 tile[kTileD * NBP];                 // exponent records: [column group of 32][K block][column]
    __shared__ uint4 tileQ[kGroups * S::kQRecs];         // weighted-sum A operands: [group][piece][instruction][lane = 32 h + c] x 8 f16

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, b, split;
    workgroup_coords(sp, bx, b, split);
    const int ns = sp.n_splits;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int rec0 = half * 32 + l31;
    const float qs = __builtin_sqrtf(prm.s2);            // sqrt(s): on both sides of the exponent products and on the q components

    int row_begin, row_end, q_begin, q_end;
    block_extent<SPARSE>(rg, N, kRowsPerBlock, row_begin, row_end, q_begin, q_end, bx);

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // rows c > D of the A operands are never written again: zero them once
    for (int t = tid; t < kGroups * S::kQRecs; t += kThreads) tileQ[t] = uint4{0u, 0u, 0u, 0u};

    for (int row0 = row_begin; row0 < row_end; row0 += kRowsPerBlock) {
        float centre[D];
        load_point<D, T>(prm.x, (long)b * N + row0, centre);

        const int wave_row0 = row0 + wave * 32;
        const bool wave_active = wave_row0 < row_end;
        uint4 X[NM];
        f32x16 acc = zero16;
        {
            const int i = min(wave_row0 + l31, row_end - 1);
            float xi[D];
            load_point<D, T>(prm.x, (long)b * N + i, xi);
            float a[D], n2 = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float xt = xi[d] - centre[d];
                n2 = __builtin_fmaf(xt, xt, n2);
                a[d] = xt * qs;
            }
            // C_i = r_i - LSE2_i (+ the margin of the value-and-gradient mode) + the 2^13 of the weights
            float cst = -0.5f * prm.s2 * n2 - (prm.fwd[(long)b * N + i] / prm.out_scale + prm.tscale) + (float)kWqShift;
            cst = __builtin_fminf(__builtin_fmaxf(cst, kH2Floor), -kH2Floor);
#pragma unroll
            for (int mm = 0; mm < NM; ++mm)
                X[mm] = select_u4(half != 0, xd_record_of<D, true, L>(2 * mm + 1, cst, a), xd_record_of<D, true, L>(2 * mm, cst, a));
        }

        for (int q = q_begin + (SPARSE ? split : 0); q < q_end; q += (SPARSE ? ns : 1)) {
            int js, je;
            column_interval<SPARSE>(rg, M, q, split, ns, js, je);
            for (int j0 = js; j0 < je; j0 += kTileD) {
                const int n = min(kTileD, je - j0);
                const int npad = (n + 31) & ~31;
                __syncthreads();
                for (int t = tid; t < npad; t += kThreads) {
                    float yt[D], H = kNegBig;
#pragma unroll
                    for (int d = 0; d < D; ++d) yt[d] = 0.f;
                    if (t < n) {
                        const long col = (long)b * M + j0 + t;
                        float yj[D];
                        load_point<D, T>(prm.y, col, yj);
                        float n2 = 0.f;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            yt[d] = yj[d] - centre[d];
                            n2 = __builtin_fmaf(yt[d], yt[d], n2);
                        }
                        H = __builtin_fmaf(-0.5f * prm.s2, n2, prm.s[col] * kLog2e);
                    }
                    H = __builtin_fmaxf(H, kH2Floor);
#pragma unroll
                    for (int d = 0; d < D; ++d) yt[d] *= qs;
                    uint4* base = &tile[(t >> 5) * (32 * NBP) + (t & 31)];
#pragma unroll
                    for (int r = 0; r < NBP; ++r) base[r * 32] = xd_record_of<D, false, L>(r, H, yt);
                    // the column's place in the weighted-sum operands: column jj of its group is register r = 4 (jj / 8) + jj % 4 of
                    // lane half (jj % 8) / 4, i.e. K slot r % 8 of instruction r / 8
                    const int jj = t & 31;
                    const int r = ((jj >> 3) << 2) | (jj & 3), hq = (jj >> 2) & 1;
                    unsigned short* qb = reinterpret_cast<unsigned short*>(&tileQ[(t >> 5) * S::kQRecs + (r >> 3) * 64 + hq * 32]) + (r & 7);
#pragma unroll
                    for (int c = 0; c <= D; ++c) {       // element [piece][instruction][lane 32 hq + c][slot]: 8 shorts per lane, 128 records per piece
                        uint32_t pc[2];
                        split2_h(c < D ? yt[c < D ? c : 0] : (t < n ? 1.f : 0.f), pc);
                        qb[c * 8] = (unsigned short)pc[0];
                        qb[128 * 8 + c * 8] = (unsigned short)pc[1];
                    }
                }
                __syncthreads();
                if (!wave_active) continue;

                for (int G = 0; G < npad / 32; ++G) {
                    const f32x16 u = xd_block<NM, NBP, L>(&tile[G * (32 * NBP)], rec0, X, zero16);
                    Pack16h whi[2], wlo[2];
#pragma unroll
                    for (int k = 0; k < 16; k += 2) {      // pairs: one v_cvt_pk_f16_f32 per two high pieces, both read back from it
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
                        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                        const f32x2_t w = {fast_exp2(u[k]), fast_exp2(u[k + 1])};
                        const f16x2_t h = __builtin_convertvector(w, f16x2_t);
                        const f32x2_t back = __builtin_convertvector(h, f32x2_t);
                        const f16x2_t l = __builtin_convertvector(w - back, f16x2_t);
                        whi[k >> 3].v[k & 7] = h[0];
                        whi[k >> 3].v[(k & 7) + 1] = h[1];
                        wlo[k >> 3].v[k & 7] = l[0];
                        wlo[k >> 3].v[(k & 7) + 1] = l[1];
                    }
                    const uint4* qg = &tileQ[G * S::kQRecs + lane];
                    Pack16h qh0, qh1, ql0, ql1;
                    qh0.u = qg[0]; qh1.u = qg[64]; ql0.u = qg[128]; ql1.u = qg[192];
                    f32x16 t = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh0.v, whi[0].v, zero16, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh1.v, whi[1].v, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql0.v, whi[0].v, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql1.v, whi[1].v, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh0.v, wlo[0].v, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh1.v, wlo[1].v, t, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((r & 3) + 8 * (r >> 2) <= D) acc[r] += t[r];      // registers whose component (of either lane half) exists
                }
            }
        }

        if (wave_active) {
            // component c of row i lives in register r(c) = (c & 3) + 4 (c >> 3) of lane half (c >> 2) & 1; the mass (c = D) goes to both
            constexpr int rD = (D & 3) + 4 * (D >> 3), hD = (D >> 2) & 1;
            float mass = acc[rD];
            mass = __shfl(mass, hD * 32 + l31, 64) * (1.0f / (float)(1 << kWqShift));
            const float unscale = 1.0f / ((float)(1 << kWqShift) * qs);
            const int i = wave_row0 + l31;
            if (i < row_end) {
                const long idx = (long)b * N + i;
                float xi[D];
                load_point<D, T>(prm.x, idx, xi);
                float* part = sp.workspace + split * sp.split_stride + idx * (D + 1);
                const float gi = (ns == 1 && prm.g) ? prm.g[idx] : 1.f;
                const float inv = (mass > 0.f) ? 1.0f / mass : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    constexpr int kD1 = D - 1;
                    const int c0 = (r & 3) + 8 * (r >> 2);        // compile-time after unrolling; the lane half adds 4
                    if (c0 <= D) {
                        const int ca = c0 < D ? c0 : kD1, cb = c0 + 4 < D ? c0 + 4 : kD1;      // (register arrays: constant indices only)
                        const float xt = half ? xi[cb] - centre[cb] : xi[ca] - centre[ca];
                        const int c = c0 + 4 * half;
                        if (c < D) {
                            const float sy = acc[r] * unscale;       // sum_j P_ij yt_j[c]
                            if (ns == 1) prm.gx[idx * D + c] = gi * (xt - sy * inv);
                            else part[c] = sy;
                        }
                    }
                }
                if (half == 0) {
                    if (ns == 1) {
                        if (prm.out) prm.out[idx] = prm.fwd[idx] + prm.out_scale * (prm.tscale + fast_log2(mass));
                    } else {
                        part[D] = mass;
                    }
                }
            }
        }
    }
}

}  // namespace glhip
