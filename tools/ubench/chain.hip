// Micro-benchmark: NM chained v_mfma_f32_32x32x16_bf16 (one 32 x 32 block of exponents for clouds of dimension 4..16) next to the
// 16 v_exp_f32 + 16 v_add_f32 of a block, on gfx950.  What does the D >= 4 soft-min loop cost per 1024 pairs, and why do the two
// pipes not overlap there (round 3: 339 cycles at NM = 5, 628 at NM = 9, i.e. matrix pipe + VALU)?
//   MODE 0: the chain alone (dependent accumulator)                         -> matrix-pipe time per chain
//   MODE 1: chain, then the 32 VALU instructions on its result (the shipped order)
//   MODE 2: two accumulator sets: chain of block k+1 interleaved with the VALU work of block k (order pinned)
//   MODE 3: as 1, but the NM MFMAs are independent (NM accumulators, summed with the exps skipped): is it the dependency?
//   MODE 4: as 2 with the y-side operands re-read from LDS (ds_read_b128 per MFMA)
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -Xclang -target-feature -Xclang -packed-fp32-ops chain.hip -o chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int M, int NM, bool CONSUME>
struct Pin {
    static __device__ __forceinline__ void run() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (CONSUME) {
            constexpr int n = (16 + NM - 1 - M) / NM;
            __builtin_amdgcn_sched_group_barrier(0x400, n, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, n, 0);
        }
        if constexpr (M + 1 < NM) Pin<M + 1, NM, CONSUME>::run();
    }
};

template <int NM, int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    __shared__ uint4 lds[64 * 16];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 16; i += 256) lds[i] = uint4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    bf16x8 a[NM], b[NM];
    for (int m = 0; m < NM; ++m)
        for (int i = 0; i < 8; ++i) { a[m][i] = (short)(lane + i + m); b[m][i] = (short)(lane * 3 + i + m); }
    f32x16 zero;
    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
    float s[4] = {seed, seed, seed, seed};
    f32x16 ua = zero, ub = zero;
    auto chain = [&](f32x16& u, int it) {
        a[0][0] = (short)it;
        u = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], zero, 0, 0, 0);
#pragma unroll
        for (int m = 1; m < NM; ++m) u = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[m], u, 0, 0, 0);
    };
    auto chain_lds = [&](f32x16& u, int it) {
        const uint4* p = &lds[(it & 1) * 8 + lane];
        union { uint4 q; bf16x8 v; } t;
        t.q = p[0];
        u = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.v, b[0], zero, 0, 0, 0);
#pragma unroll
        for (int m = 1; m < NM; ++m) { t.q = p[m * 64]; u = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t.v, b[m], u, 0, 0, 0); }
    };
    auto consume = [&](const f32x16& u) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i & 3] += __builtin_amdgcn_exp2f(u[i]);
    };
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) { chain(ua, it); s[0] += ua[0]; }
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it) { chain(ua, it); consume(ua); }
    } else if (MODE == 2 || MODE == 4) {
        if (MODE == 2) chain(ua, -1); else chain_lds(ua, -1);
        for (int it = 0; it < iters; it += 2) {
            if (MODE == 2) chain(ub, it); else chain_lds(ub, it);
            consume(ua);
            if (MODE == 4) __builtin_amdgcn_sched_group_barrier(0x100, NM, 0);
            Pin<0, NM, true>::run();
            if (MODE == 2) chain(ua, it + 1); else chain_lds(ua, it + 1);
            consume(ub);
            if (MODE == 4) __builtin_amdgcn_sched_group_barrier(0x100, NM, 0);
            Pin<0, NM, true>::run();
        }
        consume(ua);
    } else {
        f32x16 u[NM];
        for (int m = 0; m < NM; ++m) u[m] = zero;
        for (int it = 0; it < iters; ++it) {
            a[0][0] = (short)it;
#pragma unroll
            for (int m = 0; m < NM; ++m) u[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[m], u[m], 0, 0, 0);
            consume(u[0]);
        }
        for (int m = 1; m < NM; ++m) s[1] += u[m][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]) + ua[1] + ub[2];
}

static float* g_out = nullptr;

template <int NM, int MODE> void run(int wps, int iters) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL((k<NM, MODE>), dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<NM, MODE>), dim3(blocks), dim3(256), 0, 0, g_out, iters, 0.5f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("  NM=%d mode=%d : %7.1f cycles per block (1024 pairs) per SIMD (wall @ 2.4 GHz), matrix pipe alone = %d\n", NM, MODE,
           ms * 2.4e6 / ((double)iters * wps), 32 * NM);
}

template <int NM> void all(int wps, int iters) {
    run<NM, 0>(wps, iters); run<NM, 1>(wps, iters); run<NM, 2>(wps, iters); run<NM, 3>(wps, iters); run<NM, 4>(wps, iters);
}

int main() {
    (void)hipMalloc(&g_out, 256 * 8 * 256 * sizeof(float));
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {
        printf("waves per SIMD = %d\n", wps);
        if (getenv("CHAIN_SHORT")) { run<1, 1>(wps, iters); run<2, 1>(wps, iters); run<3, 1>(wps, iters); run<4, 1>(wps, iters); run<7, 1>(wps, iters); continue; }
        all<2>(wps, iters); all<3>(wps, iters); all<5>(wps, iters); all<9>(wps, iters);
    }
    return 0;
}
