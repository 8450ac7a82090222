// Micro-benchmark: do v_mfma (fp32 / bf16) and VALU transcendental work overlap on one SIMD of gfx950?
// Each wave runs ITER iterations of: NM independent MFMAs and/or NE independent v_exp_f32 (+ adds).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // bit0: f32 mfma, bit1: exp work, bit2: bf16 mfma instead of f32, bit3: plain fma work instead of exp
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    float a = seed * lane, b = seed + lane;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(lane + i); bb[i] = (short)(lane * 3 + i); }
    float e[16];
    for (int i = 0; i < 16; ++i) e[i] = seed * (i + 1) * 1e-3f - lane * 1e-4f;
    float s0 = 0.f, s1 = 0.f;
    // bit4: wave specialisation — even waves of the workgroup run only the MFMA part, odd waves only the VALU part
    const bool do_m = !(MODE & 16) || (((threadIdx.x >> 6) & 1) == 0);
    const bool do_v = !(MODE & 16) || (((threadIdx.x >> 6) & 1) == 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if ((MODE & 1) && do_m) {
                if (MODE & 4) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[g], 0, 0, 0);
                else acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
            }
            if ((MODE & 2) && do_v) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = e[g * 4 + i];
                    if (MODE & 8) v = __builtin_fmaf(v, 0.999f, 1e-6f);
                    else v = __builtin_amdgcn_exp2f(v) - 1.0f;
                    e[g * 4 + i] = v;
                }
            }
        }
    }
    for (int i = 0; i < 16; ++i) s0 += e[i];
    for (int i = 0; i < 4; ++i) s1 += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s0 + s1;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

static unsigned long long* g_clk = nullptr;
template <int MODE> float run(float* d, int blocks, int iters) {
    if (!g_clk) hipMalloc(&g_clk, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, g_clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, g_clk);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, g_clk, 16, hipMemcpyDeviceToHost);
    // clock64 = s_memtime (shader-clock domain), wall_clock64 = s_memrealtime (constant 100 MHz)
    printf("    [mode %2d] %8.3f ms | workgroup 0: clock64 %llu ticks, wall %llu ticks (%.3f ms) -> clock64 rate %.1f MHz\n", MODE, ms,
           h[0], h[1], h[1] / 1e5, h[0] / (h[1] / 100.0));
    return ms;
}

int main(int argc, char** argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 1;           // waves per SIMD
    int blocks = 256 * wps, iters = 20000;
    float* d; hipMalloc(&d, blocks * 256 * sizeof(float));
    // per iteration per wave: 4 MFMA and/or 16 exp (+16 sub)
    float t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters), t3 = run<3>(d, blocks, iters);
    float t5 = run<5>(d, blocks, iters), t7 = run<7>(d, blocks, iters);
    float t10 = run<10>(d, blocks, iters), t11 = run<11>(d, blocks, iters), t15 = run<15>(d, blocks, iters);
    float t19 = run<19>(d, blocks, iters), t23 = run<23>(d, blocks, iters), t31 = run<31>(d, blocks, iters);
    double cyc = 2.4e6;   // cycles per ms at 2.4 GHz (nominal)
    auto per = [&](float ms) { return ms * cyc / iters / wps; };   // SIMD cycles per iteration-wave
    printf("waves/SIMD=%d  (cycles per iteration per wave at nominal 2.4 GHz)\n", wps);
    printf("  f32 mfma x4 only        : %7.1f\n", per(t1));
    printf("  exp x16 (+sub) only     : %7.1f\n", per(t2));
    printf("  f32 mfma x4 + exp x16   : %7.1f   (sum %.1f, max %.1f)\n", per(t3), per(t1) + per(t2), per(t1) > per(t2) ? per(t1) : per(t2));
    printf("  bf16 mfma x4 only       : %7.1f\n", per(t5));
    printf("  bf16 mfma x4 + exp x16  : %7.1f   (sum %.1f)\n", per(t7), per(t5) + per(t2));
    printf("  fma x16 only            : %7.1f\n", per(t10));
    printf("  f32 mfma x4 + fma x16   : %7.1f   (sum %.1f)\n", per(t11), per(t1) + per(t10));
    printf("  bf16 mfma x4 + fma x16  : %7.1f   (sum %.1f)\n", per(t15), per(t5) + per(t10));
    printf("  SPECIALISED waves (half MFMA-only, half VALU-only; cycles per iteration per wave PAIR / 2):\n");
    printf("  f32 mfma | exp          : %7.1f   (each half alone: %.1f / %.1f)\n", per(t19), per(t1) / 2, per(t2) / 2);
    printf("  bf16 mfma | exp         : %7.1f   (each half alone: %.1f / %.1f)\n", per(t23), per(t5) / 2, per(t2) / 2);
    printf("  bf16 mfma | fma         : %7.1f   (each half alone: %.1f / %.1f)\n", per(t31), per(t5) / 2, per(t10) / 2);
    return 0;
}
