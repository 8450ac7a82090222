// glhip_api_grid.hip — C-ABI part 5: reductions along the lines of a regular grid and the dense row-wise soft-min.
#include "glhip_launch.h"
#include "glhip_lines.h"

extern "C" {

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
