// Micro-benchmark 2: does VALU / transcendental work issued BETWEEN independent bf16 MFMAs hide in the MFMA shadow on gfx950?
// One loop iteration = 4 x { 1 v_mfma_f32_16x16x32_bf16 ; KE x v_exp_f32 ; KA x v_add_f32 }, order pinned with
// sched_group_barrier.  DEP = 1: the exps consume the result of the MFMA issued one group earlier (as in the soft-min kernel).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -Xclang -target-feature -Xclang -packed-fp32-ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MF = 2 variant: one v_mfma_f32_32x32x16_bf16 (1024 outputs, 8 passes) per group
template <int KE, int KA>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float seed, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(lane + i); bb[i] = (short)(lane * 3 + i); }
    f32x16 d[2];
    for (int g = 0; g < 2; ++g) for (int i = 0; i < 16; ++i) d[g][i] = seed * (g + 1);
    float e[16], s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) e[i] = seed * (i + 1) * 1e-3f - lane * 1e-4f;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            d[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, d[g], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < KE; ++i) e[i & 15] = __builtin_amdgcn_exp2f(e[i & 15]);
#pragma unroll
            for (int i = 0; i < KA; ++i) s[i & 3] += e[i & 15];
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, KE + KA + 8, 0);
        }
    }
    const unsigned long long c1 = clock64();
    float r = 0.f;
    for (int g = 0; g < 2; ++g) for (int i = 0; i < 16; ++i) r += d[g][i];
    for (int i = 0; i < 16; ++i) r += e[i];
    out[blockIdx.x * 256 + threadIdx.x] = r + s[0] + s[1] + s[2] + s[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}

template <int KE, int KA, int DEP, int MF>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(lane + i); bb[i] = (short)(lane * 3 + i); }
    f32x4 c = f32x4{seed, seed, seed, seed};
    f32x4 d[4];
    for (int g = 0; g < 4; ++g) d[g] = c * (float)(g + 1);
    float e[4][4], s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < 4; ++g) for (int i = 0; i < 4; ++i) e[g][i] = seed * (g * 4 + i + 1) * 1e-3f - lane * 1e-4f;
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 prev = d[(g + 3) & 3];
            if (MF) d[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d[g], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < KE; ++i) {
                float x = DEP ? prev[i & 3] + (float)(i >> 2) : e[g][i & 3];
                float v = __builtin_amdgcn_exp2f(x);
                if (DEP) s[i & 3] += v; else e[g][i & 3] = v;
            }
#pragma unroll
            for (int i = 0; i < KA; ++i) s[i & 3] += e[g][i & 3];
            if (MF) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (DEP ? 2 : 1) * KE + KA + 8, 0);
        }
    }
    const unsigned long long c1 = clock64();
    float r = 0.f;
    for (int g = 0; g < 4; ++g) { r += s[g] + d[g].x + d[g].y + d[g].z + d[g].w; for (int i = 0; i < 4; ++i) r += e[g][i]; }
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}

static unsigned long long* g_clk = nullptr;
static float* g_out = nullptr;

template <int KE, int KA, int DEP, int MF> void run(int wps, int iters) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL((k<KE, KA, DEP, MF>), dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f, g_clk);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<KE, KA, DEP, MF>), dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f, g_clk);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long h; (void)hipMemcpy(&h, g_clk, 8, hipMemcpyDeviceToHost);
    // SIMD cycles per group (1 MFMA + KE exp + KA add), all resident waves of the SIMD counted: wall * 2.4 GHz / (iters*4*wps)
    printf("  mfma=%d exp=%d add=%d dep=%d : %6.1f cyc/group/SIMD (wall@2.4GHz) | wave 0: %6.1f cyc/group\n", MF, KE, KA, DEP,
           ms * 2.4e6 / (iters * 4.0 * wps), (double)h / (iters * 4.0));
}

// The inner loop of softmin_fwd_x32_kernel without its LDS traffic: a chained pair of 32x32x16 MFMAs (C = 0, then
// accumulate), then exp2 of the 16 results of the PREVIOUS pair and the 16 adds of the row sum.
__global__ void __launch_bounds__(256) kloop(float* out, int iters, float seed, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    bf16x8 ab, bb, ab2, bb2;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(lane + i); bb[i] = (short)(lane * 3 + i); ab2[i] = (short)(lane * 5 + i); bb2[i] = (short)(lane * 7 + i); }
    f32x16 zero;
    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
    float s = seed;
    for (int it = 0; it < iters; ++it) {
        ab[0] = (short)it;   // keep the MFMAs inside the loop
        f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, zero, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab2, bb2, d, 0, 0, 0);
        float e[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(d[i]);
        s += (((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]))) + (((e[8] + e[9]) + (e[10] + e[11])) + ((e[12] + e[13]) + (e[14] + e[15])));
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// The inner loop of wsum_x32_kernel<WS_SOFTMIN_BWD> without LDS: chained MFMA pair, 16 exp2, 48 fmac + 16 add into 64 accumulators.
template <int MF>
__global__ void __launch_bounds__(256) kloop_w(float* out, int iters, float seed, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    bf16x8 ab, bb, ab2, bb2;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(lane + i); bb[i] = (short)(lane * 3 + i); ab2[i] = (short)(lane * 5 + i); bb2[i] = (short)(lane * 7 + i); }
    f32x16 zero, d;
    for (int i = 0; i < 16; ++i) { zero[i] = 0.f; d[i] = seed * i; }
    float acc[16][4];
    for (int v = 0; v < 16; ++v) for (int c = 0; c < 4; ++c) acc[v][c] = 0.f;
    float q0 = seed * lane, q1 = seed + lane, q2 = seed - lane;
    for (int it = 0; it < iters; ++it) {
        ab[0] = (short)it;
        if (MF) {
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, zero, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab2, bb2, d, 0, 0, 0);
        } else {
            for (int i = 0; i < 16; ++i) d[i] = d[i] * 0.5f;   // stand-in producer (16 VALU)
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float w = __builtin_amdgcn_exp2f(d[v]);
            acc[v][0] = __builtin_fmaf(w, q0, acc[v][0]);
            acc[v][1] = __builtin_fmaf(w, q1, acc[v][1]);
            acc[v][2] = __builtin_fmaf(w, q2, acc[v][2]);
            acc[v][3] += w;
        }
    }
    float s = 0.f;
    for (int v = 0; v < 16; ++v) for (int c = 0; c < 4; ++c) s += acc[v][c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MF> void runloop_w(int wps, int iters) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(kloop_w<MF>, dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f, g_clk);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kloop_w<MF>, dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f, g_clk);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("  wsum loop (mfma pair=%d, 16 exp + 48 fmac + 16 add, 64 accumulators) : %6.1f cyc/1024 pairs/SIMD = %5.2f cyc per 64 pairs\n",
           MF, ms * 2.4e6 / ((double)iters * wps), ms * 2.4e6 / ((double)iters * wps) / 16.0);
}

void runloop(int wps, int iters) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(kloop, dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f, g_clk);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kloop, dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f, g_clk);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("  x32 loop (2 chained mfma32x32x16 + 16 exp + 16 add) : %6.1f cyc/1024 pairs/SIMD = %5.2f cyc per 64 pairs (wall@2.4GHz)\n",
           ms * 2.4e6 / ((double)iters * wps), ms * 2.4e6 / ((double)iters * wps) / 16.0);
}

template <int KE, int KA> void run32(int wps, int iters) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL((k32<KE, KA>), dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f, g_clk);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k32<KE, KA>), dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f, g_clk);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("  mfma32x32x16 exp=%2d add=%2d : %6.1f cyc/group/SIMD (wall@2.4GHz)\n", KE, KA, ms * 2.4e6 / (iters * 2.0 * wps));
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    const int iters = 20000;
    (void)hipMalloc(&g_clk, 16);
    (void)hipMalloc(&g_out, (size_t)256 * 8 * 256 * sizeof(float));
    printf("waves/SIMD = %d\n", wps);
    run<0, 0, 0, 1>(wps, iters);
    run<1, 0, 0, 0>(wps, iters); run<2, 0, 0, 0>(wps, iters); run<4, 0, 0, 0>(wps, iters);
    run<0, 4, 0, 0>(wps, iters);
    run<1, 0, 0, 1>(wps, iters); run<2, 0, 0, 1>(wps, iters); run<3, 0, 0, 1>(wps, iters); run<4, 0, 0, 1>(wps, iters);
    run<0, 2, 0, 1>(wps, iters); run<0, 4, 0, 1>(wps, iters); run<0, 8, 0, 1>(wps, iters);
    run<4, 4, 0, 1>(wps, iters);
    run<4, 0, 1, 1>(wps, iters); run<4, 0, 1, 0>(wps, iters);
    run32<0, 0>(wps, iters); run32<4, 0>(wps, iters); run32<8, 0>(wps, iters); run32<16, 0>(wps, iters);
    runloop(wps, iters);
    runloop_w<1>(wps, iters); runloop_w<0>(wps, iters);
    run32<0, 8>(wps, iters); run32<0, 16>(wps, iters); run32<8, 8>(wps, iters); run32<16, 16>(wps, iters);
    return 0;
}
