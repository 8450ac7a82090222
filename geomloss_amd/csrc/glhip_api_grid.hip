// glhip_api_grid.hip — C-ABI part 5: reductions along the lines of a regular grid and the dense row-wise soft-min.
#include "glhip_launch.h"
#include "glhip_lines.h"

namespace {

// ---- the elementwise front and back end of a Sinkhorn loss on a few thousand points (round 6): such a loss is bound by the host's
// launch rate (26 launches, 0.39 ms at N = 2000, of which the soft-mins are 9 launches and 0.16 ms), so the three multi-tensor
// launches of `log_weights` and the seven of the loss formula become one kernel each.

struct LogWeightsArgs {
    const float* w[4];
    float* out[4];
    long n[4];
    int count;
};

// log(w) with log(0) -> -100000 (sinkhorn_divergence.py:61-65): max(log(max(w, 0)), -100000); NaN stays NaN
__global__ void __launch_bounds__(256) log_weights_kernel(LogWeightsArgs a) {
    const int k = blockIdx.y;
    if (k >= a.count) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n[k]; i += (long)gridDim.x * 256) {
        const float w = a.w[k][i];
        const float l = logf(w > 0.f ? w : (w == w ? 0.f : w));      // (w <= 0 -> log 0 = -inf; NaN propagates)
        a.out[k][i] = (l == l) ? fmaxf(l, -100000.0f) : l;
    }
}

// out[b] = sum_i a_i (f_ba_i - f_aa_i) + sum_j b_j (g_ab_j - g_bb_j), accumulated in float64 in a fixed order (one workgroup per
// batch item, thread t takes elements t, t + 1024, ...; butterfly + LDS): sinkhorn_cost, balanced case, sinkhorn_divergence.py:171-199
__global__ void __launch_bounds__(1024) sinkhorn_cost_kernel(const float* __restrict__ a, const float* __restrict__ f_ba,
                                                             const float* __restrict__ f_aa, const float* __restrict__ b,
                                                             const float* __restrict__ g_ab, const float* __restrict__ g_bb,
                                                             float* __restrict__ out, int N, int M, long a_stride, long b_stride) {
    __shared__ double part[16];
    const int bi = blockIdx.x, tid = threadIdx.x;
    const float* ab = a + (long)bi * a_stride;      // stride 0: one weight vector shared by the batch
    const float* bb = b + (long)bi * b_stride;
    double acc = 0.0;
    for (int i = tid; i < N; i += 1024) {
        const long k = (long)bi * N + i;
        acc += (double)ab[i] * ((double)f_ba[k] - (f_aa ? (double)f_aa[k] : 0.0));
    }
    for (int j = tid; j < M; j += 1024) {
        const long k = (long)bi * M + j;
        acc += (double)bb[j] * ((double)g_ab[k] - (g_bb ? (double)g_bb[k] : 0.0));
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += part[w];
        out[bi] = (float)tot;
    }
}

// lo_hi[d] = min, lo_hi[D + d] = max of coordinate d over the rows of x and y together: one workgroup, no atomics (exact whatever
// the order): the front end of a loss whose `diameter` is measured (max_diameter, sinkhorn_divergence.py:96-112)
template <typename T>
__global__ void __launch_bounds__(1024) bounding_box_kernel(const T* __restrict__ x, long nx, const T* __restrict__ y, long ny, int D,
                                                            float* __restrict__ lo_hi) {
    const int tid = threadIdx.x;
    // thread t owns coordinate t % D of the points t / D, t / D + stride, ... (stride = the points a pass of the workgroup covers)
    const int per = 1024 / D, d = tid % D, slot = tid / D;
    float lo = INFINITY, hi = -INFINITY;
    bool nan = false;      // torch.aminmax propagates NaN coordinates (fminf / fmaxf drop them)
    if (slot < per) {
        for (long i = slot; i < nx; i += per) { const float v = to_f32<T>(x[i * D + d]); lo = fminf(lo, v); hi = fmaxf(hi, v); nan |= v != v; }
        for (long i = slot; i < ny; i += per) { const float v = to_f32<T>(y[i * D + d]); lo = fminf(lo, v); hi = fmaxf(hi, v); nan |= v != v; }
    }
    if (nan) lo = hi = NAN;
    // the per-thread extrema of a coordinate meet through LDS
    __shared__ float all_lo[1024], all_hi[1024];
    all_lo[tid] = lo; all_hi[tid] = hi;
    __syncthreads();
    if (tid < D) {
        float a = INFINITY, b = -INFINITY;
        bool bad = false;
        for (int t = tid; t < per * D; t += D) { a = fminf(a, all_lo[t]); b = fmaxf(b, all_hi[t]); bad |= all_lo[t] != all_lo[t]; }
        lo_hi[tid] = bad ? NAN : a; lo_hi[D + tid] = bad ? NAN : b;
    }
}

}  // namespace

extern "C" {

int glhip_bounding_box(const void* x, long nx, const void* y, long ny, int D, int in_dtype, float* lo_hi, void* stream) {
    if (nx < 0 || ny < 0 || D < 1 || D > 16) return fail(D > 16 ? GLHIP_EUNSUPPORTED : GLHIP_EINVAL, "glhip_bounding_box: bad sizes (D <= 16)");
    if (in_dtype != GLHIP_F32 && in_dtype != GLHIP_BF16) return fail(GLHIP_EINVAL, "glhip_bounding_box: bad in_dtype %d", in_dtype);
    if (!lo_hi || (nx > 0 && !x) || (ny > 0 && !y)) return fail(GLHIP_EINVAL, "glhip_bounding_box: NULL pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (in_dtype == GLHIP_F32) hipLaunchKernelGGL(bounding_box_kernel<float>, dim3(1), dim3(1024), 0, st, static_cast<const float*>(x), nx, static_cast<const float*>(y), ny, D, lo_hi);
    else hipLaunchKernelGGL(bounding_box_kernel<bf16_t>, dim3(1), dim3(1024), 0, st, static_cast<const bf16_t*>(x), nx, static_cast<const bf16_t*>(y), ny, D, lo_hi);
    return check_launch("glhip_bounding_box");
}

int glhip_log_weights(const float* const* w, float* const* out, const long* n, int count, void* stream) {
    if (count < 0 || count > 4) return fail(GLHIP_EINVAL, "glhip_log_weights: count must be 0 ... 4 (got %d)", count);
    if (count == 0) return GLHIP_OK;
    if (!w || !out || !n) return fail(GLHIP_EINVAL, "glhip_log_weights: NULL pointer");
    LogWeightsArgs a{};
    long nmax = 0;
    for (int k = 0; k < count; ++k) {
        if (n[k] < 0 || (n[k] > 0 && (!w[k] || !out[k]))) return fail(GLHIP_EINVAL, "glhip_log_weights: bad vector %d", k);
        a.w[k] = w[k]; a.out[k] = out[k]; a.n[k] = n[k];
        nmax = n[k] > nmax ? n[k] : nmax;
    }
    a.count = count;
    if (nmax == 0) return GLHIP_OK;
    const long blocks = (nmax + 255) / 256;
    hipLaunchKernelGGL(log_weights_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096), count, 1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("glhip_log_weights");
}

int glhip_sinkhorn_cost(const float* a, const float* f_ba, const float* f_aa, const float* b, const float* g_ab, const float* g_bb,
                        float* out, int B, int N, int M, int a_batched, int b_batched, void* stream) {
    if (B < 0 || N < 0 || M < 0) return fail(GLHIP_EINVAL, "glhip_sinkhorn_cost: bad sizes");
    if (B == 0) return GLHIP_OK;
    if (!out || (N > 0 && (!a || !f_ba)) || (M > 0 && (!b || !g_ab))) return fail(GLHIP_EINVAL, "glhip_sinkhorn_cost: NULL pointer");
    if ((f_aa == nullptr) != (g_bb == nullptr)) return fail(GLHIP_EINVAL, "glhip_sinkhorn_cost: f_aa and g_bb are given together (debiasing) or not at all");
    hipLaunchKernelGGL(sinkhorn_cost_kernel, dim3(B), dim3(1024), 0, static_cast<hipStream_t>(stream), a, f_ba, f_aa, b, g_ab, g_bb, out, N, M,
                       a_batched ? (long)N : 0L, b_batched ? (long)M : 0L);
    return check_launch("glhip_sinkhorn_cost");
}

static int lines_check(const char* fn, const void* a, const void* b, long R, int N, float eps, int p) {
    if (R < 0 || N < 0) return fail(GLHIP_EINVAL, "%s: negative size (R=%ld, N=%d)", fn, R, N);
    if (R == 0 || N == 0) return GLHIP_OK;
    if (!a || !b) return fail(GLHIP_EINVAL, "%s: NULL pointer", fn);
    if (N > kLineMax) return fail(GLHIP_EUNSUPPORTED, "%s: lines of more than %d samples are not supported (N=%d)", fn, kLineMax, N);
    if (!(eps > 0.f)) return fail(GLHIP_EINVAL, "%s: eps must be > 0", fn);
    if (p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "%s: p must be 1 or 2 (got %d)", fn, p);
    return GLHIP_OK;
}

// coordinate spacing in base-2 units: pixels sit at i / N, rescaled as in utils.py:245-252

static float lines_step(int N, float eps, int p) {
    return p == 2 ? std::sqrt(kLog2e / (2.0f * eps)) / (float)N : kLog2e / (eps * (float)N);
}

int glhip_lse_lines_fwd(const float* h, float* out, long R, int N, float eps, int p, void* stream) {
    int rc = lines_check("glhip_lse_lines_fwd", h, out, R, N, eps, p);
    if (rc || R == 0 || N == 0) return rc;
    const unsigned grid = (unsigned)(R < 262144 ? R : 262144);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p == 2) hipLaunchKernelGGL((lse_lines_fwd_kernel<2>), dim3(grid), dim3(kBlock), 0, st, h, out, R, N, lines_step(N, eps, p));
    else hipLaunchKernelGGL((lse_lines_fwd_kernel<1>), dim3(grid), dim3(kBlock), 0, st, h, out, R, N, lines_step(N, eps, p));
    return check_launch("glhip_lse_lines_fwd");
}

int glhip_lse_lines_bwd(const float* h, const float* lse, const float* grad_out, float* grad_h, long R, int N, float eps, int p,
                        void* stream) {
    int rc = lines_check("glhip_lse_lines_bwd", h, grad_h, R, N, eps, p);
    if (rc || R == 0 || N == 0) return rc;
    if (!lse || !grad_out) return fail(GLHIP_EINVAL, "glhip_lse_lines_bwd: NULL pointer");
    const unsigned grid = (unsigned)(R < 262144 ? R : 262144);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p == 2) hipLaunchKernelGGL((lse_lines_bwd_kernel<2>), dim3(grid), dim3(kBlock), 0, st, h, lse, grad_out, grad_h, R, N, lines_step(N, eps, p));
    else hipLaunchKernelGGL((lse_lines_bwd_kernel<1>), dim3(grid), dim3(kBlock), 0, st, h, lse, grad_out, grad_h, R, N, lines_step(N, eps, p));
    return check_launch("glhip_lse_lines_bwd");
}

int glhip_max_lines_fwd(const float* g, float* out, long R, int N, float step, int p, void* stream) {
    if (R < 0 || N < 0) return fail(GLHIP_EINVAL, "glhip_max_lines_fwd: negative size (R=%ld, N=%d)", R, N);
    if (R == 0 || N == 0) return GLHIP_OK;
    if (!g || !out) return fail(GLHIP_EINVAL, "glhip_max_lines_fwd: NULL pointer");
    if (N > kLineMax) return fail(GLHIP_EUNSUPPORTED, "glhip_max_lines_fwd: lines of more than %d samples are not supported (N=%d)", kLineMax, N);
    if (!(step > 0.f)) return fail(GLHIP_EINVAL, "glhip_max_lines_fwd: step must be > 0");
    if (p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "glhip_max_lines_fwd: p must be 1 or 2 (got %d)", p);
    const unsigned grid = (unsigned)(R < 262144 ? R : 262144);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p == 2) hipLaunchKernelGGL((max_lines_kernel<2>), dim3(grid), dim3(kBlock), 0, st, g, out, R, N, step);
    else hipLaunchKernelGGL((max_lines_kernel<1>), dim3(grid), dim3(kBlock), 0, st, g, out, R, N, step);
    return check_launch("glhip_max_lines_fwd");
}

int glhip_softmin_dense_fwd(const float* C, const float* h, float* out, int B, int N, int M, float eps,
                            void* stream) {
    if (B < 0 || N < 0 || M < 0) return fail(GLHIP_EINVAL, "glhip_softmin_dense_fwd: bad sizes");
    if (B == 0 || N == 0) return GLHIP_OK;
    if (!out || ((!C || !h) && M > 0)) return fail(GLHIP_EINVAL, "glhip_softmin_dense_fwd: NULL pointer");
    if (!(eps > 0.f)) return fail(GLHIP_EINVAL, "glhip_softmin_dense_fwd: eps must be > 0");
    if (B > 65535) return fail(GLHIP_EUNSUPPORTED, "glhip_softmin_dense_fwd: B=%d exceeds the grid.y limit", B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows_per_block = (kBlock / 64) * kDenseRows;
    dim3 grid((N + rows_per_block - 1) / rows_per_block, B, 1);
    const float s2 = kLog2e / eps, out_scale = -eps * kLn2;
    const bool vec = (M % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(h)) % 16 == 0);
    if (vec) hipLaunchKernelGGL((softmin_dense_kernel<true>), grid, dim3(kBlock), 0, st, C, h, out, N, M, s2, out_scale);
    else hipLaunchKernelGGL((softmin_dense_kernel<false>), grid, dim3(kBlock), 0, st, C, h, out, N, M, s2, out_scale);
    return check_launch("glhip_softmin_dense_fwd");
}

}  // extern "C"
