// glhip_api_conv.hip — C-ABI part 3: kernel products (gaussian / laplacian / energy).
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
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags, n_ranges, N);
    rc = (in_dtype == GLHIP_F32)
             ? conv_typed<false, float>(kind, x, y, v, out, nullptr, nullptr, B, N, M, D, blur, rg, n_ranges, sc, flags, st)
             : conv_typed<false, bf16_t>(kind, x, y, v, out, nullptr, nullptr, B, N, M, D, blur, rg, n_ranges, sc, flags, st);
    return rc ? rc : check_launch("glhip_kernel_conv_fwd");
}

}  // extern "C"
