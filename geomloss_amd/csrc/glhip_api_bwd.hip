// glhip_api_bwd.hip — C-ABI part 2: the soft-min gradient and the hard C-transform of point clouds.
#include "glhip_launch.h"

extern "C" {

int glhip_softmin_bwd_x(const void* x, const void* y, const float* h, const float* out, const float* grad_out,
                        float* grad_x, int B, int N, int M, int D, float eps, int p, int in_dtype,
                        const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges,
                        void* workspace, size_t workspace_bytes, int flags, void* stream) {
    int rc = check_common("glhip_softmin_bwd_x", x, y, h, B, N, M, D, in_dtype, ranges_i, slices_i, redranges_j, n_ranges);
    if (rc) return rc;
    if (B == 0 || N == 0) return GLHIP_OK;   // nothing to write
    if (!out || !grad_out || !grad_x) return fail(GLHIP_EINVAL, "glhip_softmin_bwd_x: NULL out / grad_out / grad_x");
    if (!(eps > 0.f)) return fail(GLHIP_EINVAL, "glhip_softmin_bwd_x: eps must be > 0");
    if (p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "glhip_softmin_bwd_x: p must be 1 or 2 (got %d)", p);
    const Ranges rg{ranges_i, slices_i, redranges_j};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags, n_ranges, N);
    rc = (in_dtype == GLHIP_F32)
             ? softmin_typed<true, float>(x, y, h, nullptr, out, grad_out, grad_x, B, N, M, D, eps, p, rg, n_ranges, sc, flags, st)
             : softmin_typed<true, bf16_t>(x, y, h, nullptr, out, grad_out, grad_x, B, N, M, D, eps, p, rg, n_ranges, sc, flags, st);
    return rc ? rc : check_launch("glhip_softmin_bwd_x");
}

int glhip_softmin_fwd_grad(const void* x, const void* y, const float* h, const float* guess, float margin, float* out,
                           float* grad_unit, int B, int N, int M, int D, float eps, int p, int in_dtype, const int32_t* ranges_i,
                           const int32_t* slices_i, const int32_t* redranges_j, int n_ranges, void* workspace,
                           size_t workspace_bytes, int flags, void* stream) {
    int rc = check_common("glhip_softmin_fwd_grad", x, y, h, B, N, M, D, in_dtype, ranges_i, slices_i, redranges_j, n_ranges);
    if (rc) return rc;
    if (p != 2 || D > kXdMaxD || (flags & (GLHIP_FLAG_NO_MFMA | GLHIP_FLAG_DIRECT)))
        return fail(GLHIP_EUNSUPPORTED, "glhip_softmin_fwd_grad: only p = 2, D <= 16 on the matrix-core kernels (got p %d, D %d, flags %d): "
                                        "call glhip_softmin_fwd + glhip_softmin_bwd_x", p, D, flags);
    if (B == 0 || N == 0) return GLHIP_OK;
    if (!guess || !out || !grad_unit) return fail(GLHIP_EINVAL, "glhip_softmin_fwd_grad: NULL guess / out / grad_unit");
    if (!(eps > 0.f) || !(margin >= 0.f)) return fail(GLHIP_EINVAL, "glhip_softmin_fwd_grad: eps must be > 0 and margin >= 0");
    const Ranges rg{ranges_i, slices_i, redranges_j};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags, n_ranges, N);
    StepArgs step;
    step.shift2 = margin * kLog2e / eps;      // the guess may be this far (in units of LSE2) below the exact value's upper bound
    rc = (in_dtype == GLHIP_F32)
             ? softmin_typed<true, float>(x, y, h, out, guess, nullptr, grad_unit, B, N, M, D, eps, p, rg, n_ranges, sc, flags, st, step)
             : softmin_typed<true, bf16_t>(x, y, h, out, guess, nullptr, grad_unit, B, N, M, D, eps, p, rg, n_ranges, sc, flags, st, step);
    return rc ? rc : check_launch("glhip_softmin_fwd_grad");
}

int glhip_cmin_fwd(const void* x, const void* y, const float* g, float* out, int B, int N, int M, int D, int p, int in_dtype,
                   const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges, void* workspace,
                   size_t workspace_bytes, int flags, void* stream) {
    int rc = check_common("glhip_cmin_fwd", x, y, g, B, N, M, D, in_dtype, ranges_i, slices_i, redranges_j, n_ranges);
    if (rc) return rc;
    if (p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "glhip_cmin_fwd: p must be 1 or 2 (got %d)", p);
    if (D > 3) return fail(GLHIP_EUNSUPPORTED, "glhip_cmin_fwd: D = %d > 3 is not supported", D);
    if (B == 0 || N == 0) return GLHIP_OK;
    if (!out) return fail(GLHIP_EINVAL, "glhip_cmin_fwd: NULL out");
    const Ranges rg{ranges_i, slices_i, redranges_j};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags, n_ranges, N);
    auto run = [&](auto tag) {
        using T = decltype(tag);
        const CminParams<T> prm{static_cast<const T*>(x), static_cast<const T*>(y), g, out};
#define GL_CMIN(DD, PP) launch_mapreduce<HardMinOp<DD, PP, 2, T>>(prm, rg, n_ranges, B, N, M, sc.ws, sc.bytes, sc.allow_split, st, sc.cb)
        if (p == 2) { if (D == 1) GL_CMIN(1, 2); else if (D == 2) GL_CMIN(2, 2); else GL_CMIN(3, 2); }
        else        { if (D == 1) GL_CMIN(1, 1); else if (D == 2) GL_CMIN(2, 1); else GL_CMIN(3, 1); }
#undef GL_CMIN
    };
    if (in_dtype == GLHIP_F32) run(float{}); else run(bf16_t{});
    return check_launch("glhip_cmin_fwd");
}

}  // extern "C"
