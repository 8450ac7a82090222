// Micro-benchmark 3: the weighted-sum reductions (soft-min gradient, gaussian gradient) with the sums on the matrix cores.
// Per 32 x 32 block of pairs (one wave): 2 chained v_mfma_f32_32x32x16_bf16 (exponents, C = 0), 16 v_exp_f32, the weights
// split into two bf16 pieces (hi: v_perm_b32 of the upper halves = truncation, exact residual by v_and + v_sub, lo:
// v_cvt_pk_bf16_f32), 4 accumulating v_mfma_f32_32x32x16_bf16 (B = the lane's own 16 weights, A = per-column vectors).
// Compared with the VALU form (16 exp + 48 fma + 16 add into 64 accumulators).  Prints SIMD cycles per 64 pairs.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -Xclang -target-feature -Xclang -packed-fp32-ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
union P16 { unsigned u[4]; bf16x8 v; };

template <int NACC, int PIECES>
__global__ void __launch_bounds__(256) kmm(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    bf16x8 ab, bb, ab2, bb2, qa, qb;
    for (int i = 0; i < 8; ++i) {
        ab[i] = (short)(lane + i); bb[i] = (short)(lane * 3 + i); ab2[i] = (short)(lane * 5 + i); bb2[i] = (short)(lane * 7 + i);
        qa[i] = (short)(0x3F80 + lane + i); qb[i] = (short)(0x3F00 + lane * 2 + i);
    }
    f32x16 zero, acc[2];
    for (int i = 0; i < 16; ++i) { zero[i] = 0.f; acc[0][i] = 0.f; acc[1][i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        ab[0] = (short)it;
        f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, zero, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab2, bb2, d, 0, 0, 0);
        float w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = __builtin_amdgcn_exp2f(d[i]);
        P16 hi[2], lo[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned a = __float_as_uint(w[2 * k]), b = __float_as_uint(w[2 * k + 1]);
            hi[k >> 2].u[k & 3] = __builtin_amdgcn_perm(b, a, 0x07060302u);
            if (PIECES == 2) {
                const f32x2 r = {w[2 * k] - __uint_as_float(a & 0xFFFF0000u), w[2 * k + 1] - __uint_as_float(b & 0xFFFF0000u)};
                const bf16x2 p = __builtin_convertvector(r, bf16x2);
                unsigned u; __builtin_memcpy(&u, &p, 4);
                lo[k >> 2].u[k & 3] = u;
            }
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, hi[0].v, acc[0], 0, 0, 0);
        acc[NACC - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qb, hi[1].v, acc[NACC - 1], 0, 0, 0);
        if (PIECES == 2) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, lo[0].v, acc[0], 0, 0, 0);
            acc[NACC - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qb, lo[1].v, acc[NACC - 1], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float* g_out = nullptr;
template <int NACC, int PIECES> void run(int wps, int iters) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL((kmm<NACC, PIECES>), dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((kmm<NACC, PIECES>), dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("  waves/SIMD %d: matrix-core sums, %d accumulator set(s), %d bf16 piece(s) per weight: %6.1f cyc / 1024 pairs / SIMD = %5.2f cyc per 64 pairs\n",
           wps, NACC, PIECES, ms * 2.4e6 / ((double)iters * wps), ms * 2.4e6 / ((double)iters * wps) / 16.0);
}
int main() {
    const int iters = 20000;
    (void)hipMalloc(&g_out, (size_t)256 * 8 * 256 * sizeof(float));
    for (int wps : {1, 2, 4}) { run<2, 2>(wps, iters); run<1, 2>(wps, iters); run<2, 1>(wps, iters); }
    return 0;
}
