// glhip_klayout.h — the K layouts of an exponent assembled on the matrix cores (shared by glhip_softmin_x32.h, glhip_softmin_xd.h,
// glhip_wsum_t32.h, glhip_dist_xd.h): how a column / a row becomes 16-byte MFMA operand records.  See glhip_softmin_xd.h for the
// derivation of the bf16 x 3 layout; the f16 x 2 one is described below.
#pragma once

#include "glhip_softmin_xdl.h"

namespace glhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// K layout of an exponent.
//   XL_BF16X3 (default): every fp32 operand = three bf16 pieces, six products per coordinate (above).
//   XL_F16X2  (GLHIP_FLAG_F16X2, round 5): every coordinate = TWO f16 pieces (hi + lo, 22 significant bits, round-to-nearest,
//     subnormal pieces kept: |a - (hi + lo)| <= max(2^-23 |a|, 2^-25)), THREE products per coordinate — hi hi, hi lo, lo hi
//     ([y_hi, y_lo, y_hi] against [a_hi, a_hi, a_lo]; the dropped lo lo is <= 2^-22 |a y|) — on v_mfma_f32_32x32x16_f16, which
//     keeps subnormal inputs and accumulates like the bf16 form (tools/ubench/mfma_f16.hip, profiles/r05_ubench_mfma_f16.txt).
//     Both sides carry sqrt(s) (the bf16 layout puts all of s on the rows): f16 has 5 exponent bits, and the products only need
//     |sqrt(s) (x - c)| < 65504.  The scalar item keeps six slots, [H1,H2,H3,k,k,k] against [k,k,k,n1,n2,n3] with k = 8 and the
//     three f16 pieces of H / 8 and n / 8 (33 bits, absolute floor 8 x 2^-25 = 2.4e-7 of an exponent, range |H| < 5.2e5).
//     Chained MFMAs: ceil((3 D + 6) / 16) —
//         D      1-3  4  5  6  7  8  9  10  11  12  13  14  15  16
//         NM      1   2  2  2  2  2  3   3   3   3   3   3   4   4        (bf16 x 3:  2  2 3 3 3 4 4 5 5 5 6 6 6 7)
//     and half the LDS bytes per column.  Accuracy: the cross term is good to ~2^-21 |a y| worst case (bf16 x 3: 2^-24), on top of
//     the float32 accumulation both layouts share (ulp of the partial sums, truncated); measured next to each other in
//     profiles/r05_f16x2_accuracy.txt.  RANGE IS THE CALLER'S VOUCH: exponents H_j, n_i beyond +-5e5 (log2 units), i.e. roughly
//     (cloud diameter)^2 / eps > 3e5, overflow f16 — results are then inf / nan, loudly; the Python layer sets the flag from
//     eps and the diameter it already knows.
enum XdLayout { XL_BF16X3 = 0, XL_F16X2 = 1 };
constexpr float kH2Floor = -5.0e5f;      // XL_F16X2: "minus infinity" of an exponent (padded / massless columns, rows that have seen none)
constexpr float kH2Kappa = 8.0f;         // scale of the scalar item
constexpr uint32_t kF16Kappa = 0x4800u;  // 8.0 as f16

template <int D, int L = XL_BF16X3>
struct XdShape {
    static_assert(D >= 1 && D <= 16, "glhip_softmin_xd.h serves D <= 16");
    static constexpr int kPer = (L == XL_F16X2) ? 3 : 6;     // K slots per coordinate
    static constexpr int kSlots = 6 + kPer * D;      // scalar item (slots 0..5), then kPer slots per coordinate
    static constexpr int NM = (kSlots + 15) / 16;    // chained MFMAs per 32 x 32 block
    static constexpr int NBP = 2 * NM;               // records (8 slots, 16 bytes) per column in LDS: record r = slots [8 r, 8 r + 8)
    static constexpr int kTile = NBP <= 6 ? 512 : (NBP <= 10 ? 256 : 128);   // columns per LDS tile (on-the-fly staging)
    // pre-packed columns: two tile buffers of at most 2304 records (36 KiB) each — two 8-wave workgroups per CU hold 144 of the
    // 160 KiB — in whole groups of 32 columns
    static constexpr int kTilePre = (2304 / NBP) / 32 * 32;
    static constexpr int kGroupRecs = 32 * NBP;      // records of one column group = NM chunks of 64 records (1 KiB)
};

// fp32 -> three bf16 pieces whose sum is exact, split with round-to-nearest-even (the residuals are signed and at most half a
// unit of the piece before them).  inf / nan stay in the first piece.
__device__ __forceinline__ void split3_rn(float v, uint32_t (&p)[3]) {
    const uint32_t u = __float_as_uint(v);
    const bool special = (u & 0x7F800000u) == 0x7F800000u;
    const uint32_t b1 = special ? (u & 0xFFFF0000u) : ((u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
    const float r = special ? 0.f : v - __uint_as_float(b1);                      // exact
    const uint32_t ur = __float_as_uint(r);
    const uint32_t b2 = (ur + 0x7FFFu + ((ur >> 16) & 1u)) & 0xFFFF0000u;
    const float r2 = r - __uint_as_float(b2);                                      // exact, at most 8 significant bits
    p[0] = b1 >> 16;
    p[1] = b2 >> 16;
    p[2] = __float_as_uint(r2) >> 16;
}

constexpr uint32_t kBf16One = 0x3F80u;

// ---- XL_F16X2 pieces ----
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union Pack16h { uint4 u; f16x8 v; };
__device__ __forceinline__ uint32_t f16_bits(_Float16 h) { return (uint32_t)__builtin_bit_cast(unsigned short, h); }
// v = hi + lo (+ rest <= max(2^-23 |v|, 2^-25)); inf / nan (and |v| >= 65520, which f16 rounds to inf) stay in hi alone
__device__ __forceinline__ void split2_h(float v, uint32_t (&p)[2]) {
    const _Float16 hi = (_Float16)v;
    const float hf = (float)hi;
    const float r = (__builtin_fabsf(hf) <= 65504.f) ? v - hf : 0.f;      // exact
    p[0] = f16_bits(hi);
    p[1] = f16_bits((_Float16)r);
}
__device__ __forceinline__ void split3_h(float v, uint32_t (&p)[3]) {
    const _Float16 hi = (_Float16)v;
    const float hf = (float)hi;
    const float r = (__builtin_fabsf(hf) <= 65504.f) ? v - hf : 0.f;
    const _Float16 mid = (_Float16)r;
    const float r2 = r - (float)mid;                                       // exact
    p[0] = f16_bits(hi);
    p[1] = f16_bits(mid);
    p[2] = f16_bits((_Float16)r2);
}

// bf16 value of K slot `slot` of a column (y side) or of a row (x side): sc = pieces of the scalar item (H_j | n_i),
// cp[d] = pieces of coordinate d.  All indices are compile-time constants after unrolling.
template <int D, bool XSIDE>
__device__ __forceinline__ uint32_t xd_slot(int slot, const uint32_t (&sc)[3], const uint32_t (&cp)[D][3]) {
    if (slot >= 6 * (D + 1)) return 0u;
    if (slot < 6) {
        if (XSIDE) return slot < 3 ? kBf16One : sc[slot - 3];     // [1,1,1,n1,n2,n3]
        return slot < 3 ? sc[slot] : kBf16One;                    // [H1,H2,H3,1,1,1]
    }
    const int d = (slot - 6) / 6, t = (slot - 6) % 6;
    const int piece = XSIDE ? (t == 2 ? 1 : (t == 4 ? 2 : (t == 5 ? 1 : 0)))        // [a1,a1,a2,a1,a3,a2]
                            : (t == 1 ? 1 : (t == 3 ? 2 : (t == 5 ? 1 : 0)));       // [y1,y2,y1,y3,y1,y2]
    return cp[d][piece];
}

// The same from the float values themselves, splitting what the record needs when it needs it (a record touches the scalar and
// at most three coordinates): for the row pass of the kernels, where 3 (D + 1) live piece registers next to RT x NM finished
// operands pushed D = 16 over 128 VGPRs.
template <int D, bool XSIDE, int L = XL_BF16X3>
__device__ __forceinline__ uint4 xd_record_of(int r, float scalar, const float (&val)[D]) {
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if constexpr (L == XL_F16X2) {      // scalar item [k,k,k,n1,n2,n3] | [H1,H2,H3,k,k,k], then [a_hi,a_hi,a_lo] | [y_hi,y_lo,y_hi] per coordinate
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int slot = 8 * r + k;
            uint32_t h = 0u;
            if (slot < 6 + 3 * D) {
                if (slot < 6) {
                    const bool one = XSIDE ? slot < 3 : slot >= 3;
                    if (one) h = kF16Kappa;
                    else { uint32_t p[3]; split3_h(scalar * (1.0f / kH2Kappa), p); h = p[XSIDE ? slot - 3 : slot]; }
                } else {
                    const int d = (slot - 6) / 3, t = (slot - 6) % 3;
                    uint32_t p[2];
                    split2_h(val[d < D ? d : 0], p);
                    h = p[XSIDE ? (t == 2 ? 1 : 0) : (t == 1 ? 1 : 0)];
                }
            }
            w[k >> 1] |= (k & 1) ? (h << 16) : h;
        }
        return uint4{w[0], w[1], w[2], w[3]};
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int slot = 8 * r + k;
        uint32_t h = 0u;
        if (slot < 6 * (D + 1)) {
            uint32_t p[3];
            if (slot < 6) {
                const bool one = XSIDE ? slot < 3 : slot >= 3;
                if (one) h = kBf16One;
                else { split3_rn(scalar, p); h = p[XSIDE ? slot - 3 : slot]; }
            } else {
                const int d = (slot - 6) / 6, t = (slot - 6) % 6;
                const int piece = XSIDE ? (t == 2 ? 1 : (t == 4 ? 2 : (t == 5 ? 1 : 0))) : (t == 1 ? 1 : (t == 3 ? 2 : (t == 5 ? 1 : 0)));
                split3_rn(val[d < D ? d : 0], p);
                h = p[piece];
            }
        }
        w[k >> 1] |= (k & 1) ? (h << 16) : h;
    }
    return uint4{w[0], w[1], w[2], w[3]};
}

// record r (slots 8 r .. 8 r + 7) of a column / a row
template <int D, bool XSIDE>
__device__ __forceinline__ uint4 xd_record(int r, const uint32_t (&sc)[3], const uint32_t (&cp)[D][3]) {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        w[k] = xd_slot<D, XSIDE>(8 * r + 2 * k, sc, cp) | (xd_slot<D, XSIDE>(8 * r + 2 * k + 1, sc, cp) << 16);
    return uint4{w[0], w[1], w[2], w[3]};
}

// x side, record 0 of lane half 0 = [1,1,1,n1,n2,n3, coordinate 0: a1, a1]: replaces n (soft-min: minus the running maximum)
template <int L = XL_BF16X3>
__device__ __forceinline__ uint4 xd_with_n(const uint4& rec0, float n) {
    uint32_t p[3];
    if constexpr (L == XL_F16X2) {
        split3_h(__builtin_fminf(__builtin_fmaxf(n, kH2Floor), -kH2Floor) * (1.0f / kH2Kappa), p);
        return uint4{rec0.x, kF16Kappa | (p[0] << 16), p[1] | (p[2] << 16), rec0.w};
    }
    split3_rn(n, p);
    return uint4{rec0.x, kBf16One | (p[0] << 16), p[1] | (p[2] << 16), rec0.w};
}

__device__ __forceinline__ f32x16 mfma_h32(const uint4& a, const uint4& b, const f32x16& c) {
    Pack16h pa, pb;
    pa.u = a;
    pb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(pa.v, pb.v, c, 0, 0, 0);
}

}  // namespace glhip
