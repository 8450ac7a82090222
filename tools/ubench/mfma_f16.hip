// v_mfma_f32_32x32x16_f16 as the carrier of "f16 x 2" split products (round 5, lead 6 of the round-4 review): what has to hold for
// an fp32 operand a = a_hi + a_lo (two f16 pieces, 22 bits) to be usable with 3 products per coordinate instead of bf16 x 3's 6:
//   1. subnormal f16 INPUTS are not flushed (a_lo of a small coordinate is subnormal: spacing 2^-24 absolute);
//   2. the product of two 11-bit significands is exact in the fp32 accumulation;
//   3. v_cvt_f16_f32 rounds to nearest and produces subnormals (so hi + lo reproduces a to 2^-23 relative, or 2^-25 absolute);
//   4. +-inf pieces behave (H = -inf columns), and what overflow looks like (65520 -> inf).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f16.hip -o tools/ubench/mfma_f16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* av, const float* bv, float c, float* out) {
    const int lane = threadIdx.x, half = lane >> 5;
    f16x8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (_Float16)av[8 * half + t]; b[t] = (_Float16)bv[8 * half + t]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

// split / recombine on the device: |v - (hi + lo)| and the pieces
__global__ void split(const float* v, float* out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const _Float16 hi = (_Float16)v[i];
    const float r = v[i] - (float)hi;
    const _Float16 lo = (_Float16)r;
    out[3 * i] = (float)hi; out[3 * i + 1] = (float)lo; out[3 * i + 2] = r - (float)lo;
}

static void run(const char* what, const float (&a)[16], const float (&b)[16], float c, double exact) {
    float *da, *db, *dout, h;
    (void)hipMalloc(&da, 64); (void)hipMalloc(&db, 64); (void)hipMalloc(&dout, 4);
    (void)hipMemcpy(da, a, 64, hipMemcpyHostToDevice); (void)hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, c, dout);
    (void)hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
    printf("  %-84s got % .9e   exact % .9e\n", what, h, exact);
}

int main() {
    printf("tools/ubench/mfma_f16.hip (v_mfma_f32_32x32x16_f16)\n");
    const float sub = ldexpf(1.f, -20), sub2 = ldexpf(3.f, -24);     // f16 subnormals (min normal 2^-14)
    { float a[16] = {sub}, b[16] = {1024.f}; run("subnormal input 2^-20 x 1024 (flushed: 0)", a, b, 0.f, ldexp(1.0, -10)); }
    { float a[16] = {sub2}, b[16] = {2048.f}; run("smallest subnormals 3 x 2^-24 x 2048", a, b, 0.f, 3 * ldexp(1.0, -13)); }
    { float a[16] = {sub, sub}, b[16] = {sub, 1.f}; run("subnormal x subnormal (2^-40) + subnormal x 1", a, b, 0.f, ldexp(1.0, -40) + ldexp(1.0, -20)); }
    { const float q = 1.f + ldexpf(1.f, -10); float a[16] = {q}, b[16] = {q}; run("(1 + 2^-10)^2: 21-bit product, exact in fp32", a, b, 0.f, (double)q * q); }
    { const float q = 2047.f; float a[16] = {q, 1.f}, b[16] = {q, ldexpf(1.f, -3)}; run("2047^2 + 2^-3 (4190209.125: needs 25 bits; fp32 rounds / truncates)", a, b, 0.f, 2047.0 * 2047.0 + 0.125); }
    { float a[16] = {60000.f}, b[16] = {60000.f}; run("60000^2 (fits fp32)", a, b, 0.f, 3.6e9); }
    { float a[16] = {-INFINITY, 5.f}, b[16] = {1.f, 3.f}; run("-inf x 1 + 15", a, b, 0.f, -INFINITY); }
    { float a[16] = {-INFINITY, 0.f}, b[16] = {1.f, 3.f}; run("-inf x 1 + 0 x 3, C = 1e30", a, b, 1e30f, -INFINITY); }
    { float a[16] = {65520.f}, b[16] = {1.f}; run("65520 (rounds to +inf in f16; 65519 -> 65504)", a, b, 0.f, INFINITY); }
    { float a[16] = {-62500.f, 8.f}, b[16] = {8.f, 62500.f}; run("-62500 x 8 + 8 x 62500 (scaled scalar items cancel exactly)", a, b, 0.f, 0.0); }
    { float a[16] = {256.f, -256.f, sub}, b[16] = {1.f, 1.f, 1.f}; run("256 - 256 + 2^-20 inside one instruction (wide adder?)", a, b, 0.f, sub); }

    const float vals[8] = {0.123456789f, 21.7654321f, 3.3e-4f, 6.0e-5f, 1.0e-6f, 499.999f, -0.0712345f, 600.12345f};
    float *dv, *dout, h[24];
    (void)hipMalloc(&dv, 32); (void)hipMalloc(&dout, 96);
    (void)hipMemcpy(dv, vals, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split, dim3(1), dim3(64), 0, 0, dv, dout, 8);
    (void)hipMemcpy(h, dout, 96, hipMemcpyDeviceToHost);
    printf("  two-piece split v = hi + lo + rest (v_cvt_f16_f32, round to nearest, subnormal pieces kept):\n");
    for (int i = 0; i < 8; ++i)
        printf("    v % .9e  hi % .9e  lo % .9e  rest % .3e  (rest / v = %.2e, 2^-23 = 1.19e-07, 2^-25 = 2.98e-08)\n", vals[i], h[3 * i], h[3 * i + 1],
               h[3 * i + 2], fabs(h[3 * i + 2] / vals[i]));
    return 0;
}
