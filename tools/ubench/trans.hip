// Micro-benchmark: issue cost of the quarter-rate VALU instructions of gfx950 (v_exp_f32, v_sqrt_f32, v_rsq_f32, v_log_f32,
// v_rcp_f32) alone and next to the plain instruction each distance-type kernel pairs them with.
//   build: hipcc --offload-arch=gfx950 -O3 tools/ubench/trans.hip -o tools/ubench/trans ; run: tools/ubench/trans [waves per SIMD]
// One iteration of a wave = 16 independent chains of { T transcendentals, P plain ops }.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

enum { EXP = 0, SQRT = 1, RSQ = 2, LOG = 3, RCP = 4, NONE = 5 };

template <int OP> __device__ __forceinline__ float trans(float v) {
    if (OP == EXP) return __builtin_amdgcn_exp2f(v);
    if (OP == SQRT) return __builtin_amdgcn_sqrtf(v);
    if (OP == RSQ) return __builtin_amdgcn_rsqf(v);
    if (OP == LOG) return __builtin_amdgcn_logf(v);
    if (OP == RCP) return __builtin_amdgcn_rcpf(v);
    return v;
}

// OP2 != NONE: a second transcendental per chain (sqrt then exp: the laplacian / p = 1 soft-min pattern); PLAIN fma per chain
template <int OP, int OP2, int PLAIN>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float e[16], acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) e[i] = seed * (i + 1) * 1e-3f + lane * 1e-4f + 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = trans<OP>(e[i]);
            if (OP2 != NONE) v = trans<OP2>(-v);
#pragma unroll
            for (int p = 0; p < PLAIN; ++p) acc[(i + p) & 3] = __builtin_fmaf(v, 0.999f + p, acc[(i + p) & 3]);
            if (PLAIN == 0) acc[i & 3] += v * 0.f;      // keeps the chain alive without an issue slot worth mentioning
            e[i] = __builtin_fabsf(e[i]) * 0.9999f + 1e-3f * (it & 1);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]) + e[3];
}

template <int OP, int OP2, int PLAIN> double run(float* d, int blocks, int iters, int wps, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP, OP2, PLAIN>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, OP2, PLAIN>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 2.4e6 / iters / wps / 16.0;     // SIMD cycles (nominal 2.4 GHz) per chain and wave
    printf("  %-44s : %6.2f cycles per chain (64 lanes)\n", name, cyc);
    return cyc;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 8;
    const int blocks = 256 * wps, iters = 4000;
    float* d; hipMalloc(&d, (size_t)blocks * 256 * sizeof(float));
    printf("tools/ubench/trans.hip, %d waves per SIMD; every chain also carries 2 plain ops (abs-mul-add of the input refresh)\n", wps);
    const double base = run<NONE, NONE, 0>(d, blocks, iters, wps, "refresh only (2 plain ops)");
    run<NONE, NONE, 1>(d, blocks, iters, wps, "+ 1 fma");
    run<EXP, NONE, 0>(d, blocks, iters, wps, "+ v_exp_f32");
    run<SQRT, NONE, 0>(d, blocks, iters, wps, "+ v_sqrt_f32");
    run<RSQ, NONE, 0>(d, blocks, iters, wps, "+ v_rsq_f32");
    run<LOG, NONE, 0>(d, blocks, iters, wps, "+ v_log_f32");
    run<RCP, NONE, 0>(d, blocks, iters, wps, "+ v_rcp_f32");
    run<EXP, NONE, 1>(d, blocks, iters, wps, "+ v_exp_f32 + fma      (gaussian / p = 2)");
    run<SQRT, NONE, 1>(d, blocks, iters, wps, "+ v_sqrt_f32 + fma     (energy)");
    run<RSQ, NONE, 2>(d, blocks, iters, wps, "+ v_rsq_f32 + mul + fma (energy via rsq)");
    run<SQRT, EXP, 1>(d, blocks, iters, wps, "+ v_sqrt_f32 + v_exp_f32 + fma (laplacian)");
    run<SQRT, EXP, 2>(d, blocks, iters, wps, "+ v_sqrt_f32 + sub + v_exp_f32 + add (p = 1)");
    (void)base;
    return 0;
}
