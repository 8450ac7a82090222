// Micro-benchmark: the weighted-sum inner loop (soft-min gradient / gaussian gradient, D = 3) on TRANSPOSED 32x32x16 blocks with the
// per-column vectors q_j = (yt_j, v_j) as SCALAR operands.  Two row tiles per wavefront; after the exponentials one
// v_permlane32_swap per register pair makes register k of all 64 lanes hold ONE column (rows of tile A in lanes 0-31, of tile B in
// lanes 32-63), so q_j can come from SGPRs (s_load from a packed global array through the scalar cache) instead of broadcast
// ds_read_b128 — no LDS traffic for q, D + 1 accumulators per lane instead of 2 (D + 1).
//   variant 0: q from LDS as broadcast float4 (the shipped glhip_wsum_t32.h loop)     variant 1: swap + SGPR q
//   NQ = 3: soft-min gradient (3 fma + 1 add per pair)   NQ = 4: gaussian gradient (4 fma per pair)
// Prints ms per 1e12 pairs (compare: shipped 16x16x32 kernels 148 / 163 ms).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -Xclang -target-feature -Xclang -packed-fp32-ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTile = 512;   // columns per LDS tile

__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
    union { uint4 u; bf16x8 v; } A, B;
    A.u = a; B.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B.v, c, 0, 0, 0);
}

template <int VARIANT, int NQ, int NW, int RT = 2>
__global__ void __launch_bounds__(NW * 64) kern(const uint4* __restrict__ rec, const float4* __restrict__ q, float* out, int M) {
    __shared__ uint4 tile[kTile * 4];
    __shared__ __attribute__((aligned(16))) float tileQ[4 * kTile];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31, rec0 = half * 32 + l31;
    uint4 X[RT][2];
    for (int rt = 0; rt < RT; ++rt)
        for (int m = 0; m < 2; ++m) X[rt][m] = uint4{0x3c003c00u + lane + rt, 0x3c003c00u + m, 0x3b003b00u, 0x3a003a00u + blockIdx.x % 7};
    float acc[RT][4];
    for (int rt = 0; rt < RT; ++rt) for (int c = 0; c < 4; ++c) acc[rt][c] = 0.f;
    const float4 qconst = q[blockIdx.x & 1023];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < M; j0 += kTile) {
        __syncthreads();
        for (int r = tid; r < kTile * 4; r += NW * 64) tile[r] = rec[(size_t)j0 * 4 + r];
        if (VARIANT == 0)
            for (int t = tid; t < kTile; t += NW * 64) {
                const float4 v = q[j0 + t];
                tileQ[t] = v.x; tileQ[kTile + t] = v.y; tileQ[2 * kTile + t] = v.z; tileQ[3 * kTile + t] = v.w;
            }
        __syncthreads();
        for (int G = 0; G < kTile / 32; ++G) {
            const uint4 ya = tile[G * 128 + rec0], yb = tile[G * 128 + 64 + rec0];
            f32x16 w[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x16 u = mfma(ya, X[rt][0], zero16);
                u = mfma(yb, X[rt][1], u);
#pragma unroll
                for (int k = 0; k < 16; ++k) w[rt][k] = __builtin_amdgcn_exp2f(u[k]);
            }
            if (VARIANT == 0) {
                const float* qg = &tileQ[G * 32 + half * 4];
#pragma unroll
                for (int c = 0; c < NQ; ++c)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const float4 q4 = *reinterpret_cast<const float4*>(qg + c * kTile + qq * 8);
                        const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[rt][c] = __builtin_fmaf(w[rt][qq * 4 + r], qv[r], acc[rt][c]);
                    }
                if (NQ == 3) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < 16; ++k) s4[k & 3] += w[rt][k];
                        acc[rt][3] += (s4[0] + s4[1]) + (s4[2] + s4[3]);
                    }
                }
            } else {
                const float4* qg = q + j0 + G * 32;      // wave-uniform: scalar loads
                float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    float wa = w[0][k], wb = w[1][k];
                    if (VARIANT != 2) {
                        const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[0][k]), __float_as_uint(w[1][k]), false, false);
                        wa = __uint_as_float(sw.x), wb = __uint_as_float(sw.y);      // column ca | cb for the lane's own row
                    }
                    const int ca = (k >> 2) * 8 + (k & 3), cb = ca + 4;
                    float4 qa, qb;
                    if (VARIANT == 3) {         // no loads: q from four kernel-lifetime scalars
                        qa = float4{qconst.x + k, qconst.y, qconst.z, qconst.w};
                        qb = float4{qconst.y, qconst.z + k, qconst.w, qconst.x};
                    } else {
                        qa = qg[ca], qb = qg[cb];
                    }
                    acc[0][0] = __builtin_fmaf(wa, qa.x, acc[0][0]);
                    acc[0][1] = __builtin_fmaf(wa, qa.y, acc[0][1]);
                    acc[0][2] = __builtin_fmaf(wa, qa.z, acc[0][2]);
                    acc[1][0] = __builtin_fmaf(wb, qb.x, acc[1][0]);
                    acc[1][1] = __builtin_fmaf(wb, qb.y, acc[1][1]);
                    acc[1][2] = __builtin_fmaf(wb, qb.z, acc[1][2]);
                    if (NQ == 4) {
                        acc[0][3] = __builtin_fmaf(wa, qa.w, acc[0][3]);
                        acc[1][3] = __builtin_fmaf(wb, qb.w, acc[1][3]);
                    } else {
                        s4[k & 3] += wa;
                        s4[(k + 2) & 3] += wb;
                    }
                }
                if (NQ == 3) acc[0][3] += (s4[0] + s4[1]) + (s4[2] + s4[3]);
            }
        }
    }
    float r = 0.f;
    for (int rt = 0; rt < RT; ++rt) for (int c = 0; c < 4; ++c) r += acc[rt][c];
    out[(size_t)blockIdx.x * NW * 64 + tid] = r;
}

template <int VARIANT, int NQ, int NW, int RT = 2>
void run(const char* name, const uint4* rec, const float4* q, float* out, int M, int wgs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kern<VARIANT, NQ, NW, RT>), dim3(wgs), dim3(NW * 64), 0, 0, rec, q, out, M);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double pairs = (double)wgs * NW * 32 * RT * M;      // 32 RT rows per wavefront
    printf("  %-58s %8.3f ms   -> %7.1f ms per 1e12 pairs   (%.1f cycles per 64 pairs at 2.4 GHz)\n", name, ms, ms * 1e12 / pairs,
           ms * 1e-3 * 2.4e9 * 1024 / (pairs / 64));
}

int main() {
    const int M = 1 << 17, wgs = 4096;
    uint4* rec; float4* q; float* out;
    hipMalloc(&rec, (size_t)M * 4 * sizeof(uint4)); hipMalloc(&q, (size_t)M * sizeof(float4)); hipMalloc(&out, (size_t)wgs * 512 * sizeof(float));
    std::vector<uint4> hr((size_t)M * 4);
    for (size_t i = 0; i < hr.size(); ++i) hr[i] = uint4{0x3c003c00u + (unsigned)(i % 13), 0x3b003b80u, 0x3a003a00u, 0xbc00bc00u};
    std::vector<float4> hq(M);
    for (int i = 0; i < M; ++i) hq[i] = float4{0.001f * (i % 97), 0.5f, -0.25f, 1.f};
    hipMemcpy(rec, hr.data(), hr.size() * sizeof(uint4), hipMemcpyHostToDevice);
    hipMemcpy(q, hq.data(), hq.size() * sizeof(float4), hipMemcpyHostToDevice);
    printf("tools/ubench/wsum_sgpr.hip: %d workgroups x %d columns\n", wgs, M);
    run<0, 3, 4>("q from LDS (float4 broadcast), soft-min gradient, 4 waves", rec, q, out, M, wgs);
    run<1, 3, 4>("swap + SGPR q,                soft-min gradient, 4 waves", rec, q, out, M, wgs);
    run<0, 4, 4>("q from LDS (float4 broadcast), gaussian gradient, 4 waves", rec, q, out, M, wgs);
    run<1, 4, 4>("swap + SGPR q,                gaussian gradient, 4 waves", rec, q, out, M, wgs);
    run<2, 4, 4>("SGPR q, NO swap (timing only),  gaussian gradient, 4 waves", rec, q, out, M, wgs);
    run<3, 4, 4>("swap, q = constants (no loads), gaussian gradient, 4 waves", rec, q, out, M, wgs);
    run<0, 3, 8>("q from LDS (float4 broadcast), soft-min gradient, 8 waves", rec, q, out, M, wgs / 2);
    run<1, 3, 8>("swap + SGPR q,                soft-min gradient, 8 waves", rec, q, out, M, wgs / 2);
    run<1, 4, 8>("swap + SGPR q,                gaussian gradient, 8 waves", rec, q, out, M, wgs / 2);
    run<0, 3, 2, 4>("q from LDS, FOUR row tiles per wavefront, soft-min gradient, 2 waves", rec, q, out, M, wgs);
    run<0, 4, 2, 4>("q from LDS, FOUR row tiles per wavefront, gaussian gradient, 2 waves", rec, q, out, M, wgs);
    run<0, 3, 4, 4>("q from LDS, FOUR row tiles per wavefront, soft-min gradient, 4 waves", rec, q, out, M, wgs / 2);
    run<0, 3, 4, 1>("q from LDS, ONE row tile per wavefront,   soft-min gradient, 4 waves", rec, q, out, M, wgs * 2);
    return 0;
}
