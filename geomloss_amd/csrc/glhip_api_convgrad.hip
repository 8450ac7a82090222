// glhip_api_convgrad.hip — C-ABI part 4: gradients of the kernel products.
#include "glhip_autosort.h"
#include "glhip_launch.h"

extern "C" {

int glhip_kernel_conv_bwd_x(int kind, const void* x, const void* y, const float* v, const float* g, float* grad_x,
                            int B, int N, int M, int D, float blur, int in_dtype, const int32_t* ranges_i,
                            const int32_t* slices_i, const int32_t* redranges_j, int n_ranges, void* workspace,
                            size_t workspace_bytes, int flags, void* stream) {
    int rc = check_common("glhip_kernel_conv_bwd_x", x, y, v, B, N, M, D, in_dtype, ranges_i, slices_i, redranges_j, n_ranges);
    if (rc) return rc;
    if (B == 0 || N == 0) return GLHIP_OK;   // nothing to write
    if (!g || !grad_x) return fail(GLHIP_EINVAL, "glhip_kernel_conv_bwd_x: NULL g / grad_x");
    if (kind < GLHIP_GAUSSIAN || kind > GLHIP_ENERGY) return fail(GLHIP_EINVAL, "glhip_kernel_conv_bwd_x: bad kind %d", kind);
    if (kind != GLHIP_ENERGY && !(blur > 0.f)) return fail(GLHIP_EINVAL, "glhip_kernel_conv_bwd_x: blur must be > 0");
    const Ranges rg{ranges_i, slices_i, redranges_j};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags, n_ranges, N);
    rc = (in_dtype == GLHIP_F32)
             ? conv_typed<true, float>(kind, x, y, v, nullptr, g, grad_x, B, N, M, D, blur, rg, n_ranges, sc, flags, st)
             : conv_typed<true, bf16_t>(kind, x, y, v, nullptr, g, grad_x, B, N, M, D, blur, rg, n_ranges, sc, flags, st);
    return rc ? rc : check_launch("glhip_kernel_conv_bwd_x");
}

int glhip_kernel_conv_fwd_grad(int kind, const void* x, const void* y, const float* v, float* out, float* grad_unit,
                               int B, int N, int M, int D, float blur, int in_dtype, const int32_t* ranges_i,
                               const int32_t* slices_i, const int32_t* redranges_j, int n_ranges, void* workspace,
                               size_t workspace_bytes, int flags, void* stream) {
    int rc = check_common("glhip_kernel_conv_fwd_grad", x, y, v, B, N, M, D, in_dtype, ranges_i, slices_i, redranges_j, n_ranges);
    if (rc) return rc;
    if (kind < GLHIP_GAUSSIAN || kind > GLHIP_ENERGY)
        return fail(GLHIP_EINVAL, "glhip_kernel_conv_fwd_grad: unknown kernel id %d", kind);
    if (D > 3 && !(kind == GLHIP_GAUSSIAN && D <= kXdMaxD && !(flags & GLHIP_FLAG_NO_MFMA)))
        return fail(GLHIP_EUNSUPPORTED, "glhip_kernel_conv_fwd_grad: D <= 3 (gaussian on the matrix cores: D <= 16) only (got kind %d, "
                                        "D %d, flags %d): call glhip_kernel_conv_fwd + glhip_kernel_conv_bwd_x", kind, D, flags);
    if (B == 0 || N == 0) return GLHIP_OK;
    if (!out || !grad_unit) return fail(GLHIP_EINVAL, "glhip_kernel_conv_fwd_grad: NULL out / grad_unit");
    if (kind != GLHIP_ENERGY && !(blur > 0.f)) return fail(GLHIP_EINVAL, "glhip_kernel_conv_fwd_grad: blur must be > 0");
    const Ranges rg{ranges_i, slices_i, redranges_j};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (kind != GLHIP_GAUSSIAN && autosort_applies(B, N, M, D, n_ranges, flags)) {      // as glhip_kernel_conv_fwd (glhip_autosort.h)
        AutoSort a;
        const int C = (N + kSortSlab - 1) / kSortSlab;
        rc = autosort_prepare(a, x, y, N, M, D, in_dtype, workspace, workspace_bytes, glhip_workspace_bytes(1, N, M, D, C), st);
        if (rc) return rc;
        if (a.on) {
            gather_f32(v, a.perm_y, a.col0, M, st);
            rc = glhip_kernel_conv_fwd_grad(kind, a.xs, a.ys, a.col0, a.out, a.out_rows, 1, N, M, D, blur, in_dtype, a.ranges_i, a.slices_i, a.red, a.C,
                                            a.inner_ws, a.inner_bytes, flags | GLHIP_FLAG_MFMA_DIST | GLHIP_FLAG_NO_SORT, stream);
            if (rc) return rc;
            scatter_f32(a.out, a.perm_x, out, N, st);
            scatter_f32(a.out_rows, a.perm_x, grad_unit, N, st, D);
            return check_launch("glhip_kernel_conv_fwd_grad");
        }
    }
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags, n_ranges, N);
    auto run = [&](auto tag) {
        using T = decltype(tag);
        ConvParams<T> prm;
        prm.x = static_cast<const T*>(x); prm.y = static_cast<const T*>(y); prm.v = v; prm.out = out; prm.g = nullptr; prm.gx = grad_unit;
        if (kind == GLHIP_GAUSSIAN) {
            prm.t = std::sqrt(0.5f * kLog2e) / blur;
            prm.gscale = -1.0f / (prm.t * blur * blur);
            prm.clamp2 = 0.f;
            if (flags & GLHIP_FLAG_NO_MFMA) launch_conv_d<GLHIP_GAUSSIAN, 2, T>(prm, rg, n_ranges, B, N, M, D, sc, st);
            else if (D > 3) {         // 4 <= D <= 16: transposed 32x32x16 kernel (glhip_wsum_t32.h)
#define GL_XD(DD) launch_gauss_grad_t32<DD, true, T>(prm, blur, rg, n_ranges, B, N, M, sc, st)
                GLHIP_XD_DISPATCH(D, GL_XD)
#undef GL_XD
            }
            else if (D == 1) launch_gauss_fwdgrad<1, T>(prm, blur, rg, n_ranges, B, N, M, sc, st);
            else if (D == 2) launch_gauss_fwdgrad<2, T>(prm, blur, rg, n_ranges, B, N, M, sc, st);
            else launch_gauss_fwdgrad<3, T>(prm, blur, rg, n_ranges, B, N, M, sc, st);
        } else if (kind == GLHIP_LAPLACIAN) {   // same scales as conv_typed (glhip_launch.h)
            prm.t = kLog2e / blur;
            prm.gscale = -1.0f / blur;
            prm.clamp2 = 1e-8f * kLog2e * kLog2e;
            if (use_mfma_dist(flags, n_ranges, B, D)) launch_dist_grad_d<GLHIP_LAPLACIAN, DG_FWDGRAD, T>(prm, rg, n_ranges, N, M, D, sc, st);
            else launch_conv_d<GLHIP_LAPLACIAN, 2, T>(prm, rg, n_ranges, B, N, M, D, sc, st);
        } else {
            prm.t = 1.0f;
            prm.gscale = -1.0f;
            prm.clamp2 = 1e-8f;
            if (use_mfma_dist(flags, n_ranges, B, D)) launch_dist_grad_d<GLHIP_ENERGY, DG_FWDGRAD, T>(prm, rg, n_ranges, N, M, D, sc, st);
            else launch_conv_d<GLHIP_ENERGY, 2, T>(prm, rg, n_ranges, B, N, M, D, sc, st);
        }
    };
    if (in_dtype == GLHIP_F32) run(float{}); else run(bf16_t{});
    return check_launch("glhip_kernel_conv_fwd_grad");
}

}  // extern "C"
