// glhip_common.h — shared device helpers for the gfx950 map-reduce kernels.
// Written for CDNA4 only: 64-lane wavefronts, 256-thread workgroups, LDS-staged
// column tiles.  No CUDA compatibility layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/glhip.h"

namespace glhip {

constexpr int kBlock = 256;   // threads per workgroup = 4 wavefronts, one per SIMD
constexpr int kTile = 1024;   // column records staged in LDS per tile (16 KiB of float4)
constexpr int kChunk = 8;     // column records consumed per inner-loop step
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNegBig = -1.0e30f;   // "minus infinity" that survives (a - a)

// bf16 is carried as raw uint16_t: widening to fp32 is a 16-bit shift.
struct bf16_t { uint16_t bits; };

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) {
    return __uint_as_float(static_cast<uint32_t>(v.bits) << 16);
}

// raw transcendental instructions (v_exp_f32 = 2^x, v_log_f32 = log2, v_rsq_f32, v_sqrt_f32)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Column-interval description handed to every kernel (KeOps "ranges" convention, see glhip.h).
struct Ranges {
    const int32_t* ranges_i;
    const int32_t* slices_i;
    const int32_t* redranges_j;
    // optional row-chunk table built by build_row_chunks_kernel (glhip_mapreduce.h): chunks[0] = number of chunks T, then
    // T triplets (row block k, first row, end row).  NULL: one workgroup per row block (the KeOps granularity).
    const int32_t* chunks;
};

// One LDS record per column point: D coordinates (already centred / scaled) + one scalar.
template <int D> struct alignas(16) Rec { float c[4]; };            // D = 2, 3
template <> struct alignas(8) Rec<1> { float c[2]; };

template <int D> __device__ __forceinline__ float& rec_tail(Rec<D>& r) { return r.c[D]; }
template <int D> __device__ __forceinline__ float rec_tail(const Rec<D>& r) { return r.c[D]; }

template <int D, typename T>
__device__ __forceinline__ void load_point(const T* __restrict__ p, long idx, float (&o)[D]) {
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = to_f32<T>(p[idx * D + d]);
}

// The one centre of a launch whose columns are packed once for all row blocks (pre-packed records, glhip_softmin_x32.h /
// glhip_wsum_x32.h): the mean of 8 rows spread evenly over batch item b.  The float32 error of an exponent assembled on the matrix
// cores grows with |x - c| |y - c|, so the centre should sit in the middle of the cloud whatever the order of the
// points: a single row (round 2: the first one) is a CORNER of the cloud as soon as the caller's points are sorted — a grid, a
// voxel-sorted scan — and costs 2-4 x the error of a central point.  Every kernel of a launch must compute the same bits: fixed
// order of additions, a power-of-two count.
template <int D, typename T>
__device__ __forceinline__ void launch_centre(const T* __restrict__ x, int b, int N, float (&c)[D]) {
#pragma unroll
    for (int d = 0; d < D; ++d) c[d] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {       // rows (2k + 1) N / 16: the middles of 8 equal runs (N < 8: some rows twice — still a row mean)
        float p[D];
        load_point<D, T>(x, (long)b * N + (((long)(2 * k + 1) * N) >> 4), p);
#pragma unroll
        for (int d = 0; d < D; ++d) c[d] += p[d];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) c[d] *= 0.125f;
}

}  // namespace glhip
