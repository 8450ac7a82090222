// glhip_api_conv.hip — C-ABI part 3: kernel products (gaussian / laplacian / energy).
#include "glhip_autosort.h"
#include "glhip_launch.h"

extern "C" {

int glhip_kernel_conv_fwd(int kind, const void* x, const void* y, const float* v, float* out, int B, int N, int M,
                          int D, float blur, int in_dtype, const int32_t* ranges_i, const int32_t* slices_i,
                          const int32_t* redranges_j, int n_ranges, void* workspace, size_t workspace_bytes,
                          int flags, void* stream) {
    int rc = check_common("glhip_kernel_conv_fwd", x, y, v, B, N, M, D, in_dtype, ranges_i, slices_i, redranges_j, n_ranges);
    if (rc) return rc;
    if (B == 0 || N == 0) return GLHIP_OK;   // nothing to write
    if (!out) return fail(GLHIP_EINVAL, "glhip_kernel_conv_fwd: NULL out");
    if (kind < GLHIP_GAUSSIAN || kind > GLHIP_ENERGY) return fail(GLHIP_EINVAL, "glhip_kernel_conv_fwd: bad kind %d", kind);
    if (kind != GLHIP_ENERGY && !(blur > 0.f)) return fail(GLHIP_EINVAL, "glhip_kernel_conv_fwd: blur must be > 0");
    const Ranges rg{ranges_i, slices_i, redranges_j};
    hipStream_t st = static_cast<hipStream_t>(stream);
    // laplacian / energy, big dense launches: sorted clouds -> distances on the matrix cores (glhip_autosort.h).  GLHIP_FLAG_GRAD_FAMILY
    // travels with the inner launch: glhip_kernel_conv_fwd_grad sorts the same way, so the two still round alike.
    if (kind != GLHIP_GAUSSIAN && autosort_applies(B, N, M, D, n_ranges, flags)) {
        AutoSort a;
        const int C = (N + kSortSlab - 1) / kSortSlab;
        rc = autosort_prepare(a, x, y, N, M, D, in_dtype, workspace, workspace_bytes, glhip_workspace_bytes(1, N, M, D, C), st);
        if (rc) return rc;
        if (a.on) {
            gather_f32(v, a.perm_y, a.col0, M, st);
            rc = glhip_kernel_conv_fwd(kind, a.xs, a.ys, a.col0, a.out, 1, N, M, D, blur, in_dtype, a.ranges_i, a.slices_i, a.red, a.C, a.inner_ws,
                                       a.inner_bytes, flags | GLHIP_FLAG_MFMA_DIST | GLHIP_FLAG_NO_SORT, stream);
            if (rc) return rc;
            scatter_f32(a.out, a.perm_x, out, N, st);
            return check_launch("glhip_kernel_conv_fwd");
        }
    }
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags, n_ranges, N);
    rc = (in_dtype == GLHIP_F32)
             ? conv_typed<false, float>(kind, x, y, v, out, nullptr, nullptr, B, N, M, D, blur, rg, n_ranges, sc, flags, st)
             : conv_typed<false, bf16_t>(kind, x, y, v, out, nullptr, nullptr, B, N, M, D, blur, rg, n_ranges, sc, flags, st);
    return rc ? rc : check_launch("glhip_kernel_conv_fwd");
}

}  // extern "C"
