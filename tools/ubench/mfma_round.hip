// How does v_mfma_f32_32x32x16_bf16 round?  Every output = sum_k a[k] b[k] + c with the same 16 products for all (i, j).
// Large terms that cancel + one small term: an exact / wide internal adder returns the small term, a float32 accumulation
// that rounds at the size of the partial sums returns 0 (or a multiple of ulp(256) = 3e-5).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_round.hip -o tools/ubench/mfma_round
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ short bf(float v) { return (short)(__float_as_uint(v) >> 16); }

__global__ void k(const float* av, const float* bv, float c, float* out, int chained) {
    const int lane = threadIdx.x, half = lane >> 5;
    bf16x8 a, b, a2, b2;
    for (int t = 0; t < 8; ++t) {
        a[t] = bf(av[8 * half + t]); b[t] = bf(bv[8 * half + t]);
        a2[t] = bf(av[16 + 8 * half + t]); b2[t] = bf(bv[16 + 8 * half + t]);
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (chained) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

static void run(const char* what, const float (&a)[32], const float (&b)[32], float c, int chained, double exact) {
    float *da, *db, *dout, h;
    (void)hipMalloc(&da, 128); (void)hipMalloc(&db, 128); (void)hipMalloc(&dout, 4);
    (void)hipMemcpy(da, a, 128, hipMemcpyHostToDevice); (void)hipMemcpy(db, b, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, c, dout, chained);
    (void)hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
    printf("  %-86s got % .9e   exact % .9e\n", what, h, exact);
}

int main() {
    const float s = ldexpf(1.f, -20), t = ldexpf(1.f, -12);
    printf("tools/ubench/mfma_round.hip (v_mfma_f32_32x32x16_bf16; ulp(256) = %.3e)\n", ldexp(1.0, -15));
    { float a[32] = {256.f, -256.f, s}, b[32] = {1.f, 1.f, 1.f}; run("256 - 256 + 2^-20, all in K slots 0-2 (one lane half)", a, b, 0.f, 0, s); }
    { float a[32] = {256.f, s}, b[32] = {1.f, 1.f}; run("256 + 2^-20 in K slots, C = -256", a, b, -256.f, 0, s); }
    { float a[32] = {256.f, 0, 0, 0, 0, 0, 0, 0, -256.f, s}, b[32] = {1.f, 0, 0, 0, 0, 0, 0, 0, 1.f, 1.f}; run("256 in K 0-7 (lanes 0-31), -256 + 2^-20 in K 8-15 (lanes 32-63)", a, b, 0.f, 0, s); }
    { float a[32] = {256.f, s, 0, 0, 0, 0, 0, 0, -256.f}, b[32] = {1.f, 1.f, 0, 0, 0, 0, 0, 0, 1.f}; run("256 + 2^-20 in K 0-7, -256 in K 8-15", a, b, 0.f, 0, s); }
    { float a[32] = {256.f}, b[32] = {1.f}; a[16] = -256.f; b[16] = 1.f; a[17] = s; b[17] = 1.f; run("chained: MFMA 1 = 256, MFMA 2 = -256 + 2^-20", a, b, 0.f, 1, s); }
    { float a[32] = {256.f, s}, b[32] = {1.f, 1.f}; a[16] = -256.f; b[16] = 1.f; run("chained: MFMA 1 = 256 + 2^-20, MFMA 2 = -256", a, b, 0.f, 1, s); }
    { float a[32] = {3.f, 3.f, 3.f, 3.f, 3.f, 3.f, 3.f, 3.f}, b[32] = {t, t, t, t, t, t, t, t}; run("C = 256, + 8 products of 3 x 2^-12 (each below half an ulp of 256, 0.0059 together)", a, b, 256.f, 0, 256.0 + 24 * ldexp(1.0, -12)); }
    { float a[32] = {1.f, 1.f, 1.f}, b[32] = {1.5f, s, -s * 0.5f}; run("1.5 + 2^-20 - 2^-21 (rounding mode: exact 1.5 + 2^-21 is below half an ulp of 1.5)", a, b, 0.f, 0, 1.5 + ldexp(1.0, -21)); }
    { float a[32] = {1.f, 1.f}, b[32] = {1.f, ldexpf(1.f, -24) * 1.5f}; run("1 + 1.5 x 2^-24 (nearest: 1 + 2^-23; truncation: 1)", a, b, 0.f, 0, 1.0 + 1.5 * ldexp(1.0, -24)); }
    { float a[32] = {-1.f, -1.f}, b[32] = {1.f, ldexpf(1.f, -24) * 1.5f}; run("-(1 + 1.5 x 2^-24)", a, b, 0.f, 0, -(1.0 + 1.5 * ldexp(1.0, -24))); }
    return 0;
}
