// glhip_api.hip — C-ABI of libgeomloss_hip.so (include/glhip.h), part 1: version / errors / scratch size and the soft-min FORWARD family
// (glhip_softmin_fwd, glhip_sinkhorn_step, glhip_sinkhorn_iter4, glhip_sinkhorn_anneal, glhip_sinkhorn_extrapolate4).  gfx950 only.
#include "glhip_autosort.h"
#include "glhip_launch.h"

namespace glhip {
thread_local char g_err[512] = "";
}

extern "C" {

int glhip_version(void) { return GLHIP_VERSION; }

size_t glhip_workspace_bytes(int B, int N, int M, int D, int n_ranges) {
    if (B <= 0 || N <= 0 || M <= 0 || D < 1 || D > kXdMaxD) return 0;   // the generic-D kernels (D > 16) do not split
    if (D > 3) {   // 4 <= D <= 16 (glhip_softmin_xd.h): split partials of 2 floats per row (+ packed columns on big dense launches)
        const int ns128 = choose_splits(n_ranges > 0 ? n_ranges : (long)B * ((N + 127) / 128), M, n_ranges, 1L << 30);
        int nf = ns128;
        if (n_ranges == 0 && M >= 65536) {
            for (int rows = 256; rows <= 512; rows *= 2) {
                const int nx = xcd_splits((long)B * ((N + rows - 1) / rows), M, kXdSlots, 32);
                nf = nf > nx ? nf : nx;
            }
        }
        if (n_ranges == 0) {      // small dense launches of the distance kernels split further (dist_small_launch_splits, glhip_launch.h)
            const long rb = (long)B * ((N + 255) / 256);
            if (rb * nf < 512) {
                long want = (512 + rb - 1) / rb;
                want = want < M / 128 ? want : M / 128;
                want = want < 32 ? want : 32;
                nf = nf > want ? nf : (int)want;
            }
        }
        // forward partials: 2 floats per row and split; gradient kernels (glhip_wsum_t32.h, 256-row blocks): D + 1
        size_t bytes = (size_t)(nf < 2 ? 0 : nf) * (size_t)B * (size_t)N * 2 * sizeof(float);
        int ng = choose_splits(n_ranges > 0 ? n_ranges : (long)B * ((N + 255) / 256), M, n_ranges, 1L << 30);
        if (n_ranges == 0 && M >= 65536) {
            const int nx = xcd_splits((long)B * ((N + 255) / 256), M, kXdSlots, 32);
            ng = ng > nx ? ng : nx;
        }
        const size_t grad = (size_t)(ng < 2 ? 0 : ng) * (size_t)B * (size_t)N * (size_t)(D + 1) * sizeof(float);
        if (n_ranges == 0 && M >= 65536 && (double)B * N * M >= 5e8)    // forward, big dense launches: up to 32 splits (one split's records per XCD L2) + the pre-packed column records
            bytes = (size_t)32 * (size_t)B * (size_t)N * 2 * sizeof(float) + 256 + (size_t)B * (size_t)((M + 31) / 32) * 32 * (size_t)(2 * ((6 * (D + 1) + 15) / 16)) * sizeof(uint4);   // XdShape<D>::NBP records per column
        bytes = bytes > grad ? bytes : grad;
        if (n_ranges > 0) bytes += chunk_table_bytes(n_ranges, N, 64);
        return bytes;
    }
    const long row_blocks = n_ranges > 0 ? n_ranges : (long)B * ((N + 2 * kBlock - 1) / (2 * kBlock));
    const int ns = choose_splits(row_blocks, M, n_ranges, 1L << 30);
    size_t bytes = ns < 2 ? 0 : (size_t)ns * (size_t)B * (size_t)N * (size_t)(D + 1) * sizeof(float);   // widest partial: D + 1 floats
    if (n_ranges == 0 && M >= 65536) {   // the XCD-aware grids want 8 splits (widest partial: D + 1 floats per row)
        const size_t xcd = (size_t)8 * (size_t)B * (size_t)N * (size_t)(D + 1) * sizeof(float);
        bytes = bytes > xcd ? bytes : xcd;
    }
    {
        // forward: its split partials (2 floats per row) + the packed column records (64 bytes per column, glhip_softmin_x32.h)
        int nf;
        if (n_ranges > 0) {
            nf = ns;
        } else {
            const int ns128 = choose_splits((long)B * ((N + 127) / 128), M, 0, 1L << 30), ns256 = choose_splits((long)B * ((N + 255) / 256), M, 0, 1L << 30);
            nf = ns128 > ns256 ? ns128 : ns256;
            if (M >= 65536) {
                const int x128 = xcd_splits_prepacked((long)B * ((N + 127) / 128), M, kFwdSlots, 32), x256 = xcd_splits_prepacked((long)B * ((N + 255) / 256), M, kFwdSlots, 32);
                const int nx = x128 > x256 ? x128 : x256;
                nf = nf > nx ? nf : nx;
            }
        }
        const size_t fwd = (size_t)(nf < 2 ? 0 : nf) * (size_t)B * (size_t)N * 2 * sizeof(float) + 256 + (size_t)B * (size_t)((M + 31) / 32) * 2048;
        bytes = bytes > fwd ? bytes : fwd;
    }
    {
        // weighted-sum kernels (soft-min gradient, gaussian product / gradient): split partials of up to D + 1 floats per row
        // + packed records + up to 4 q components per column
        int nw = ns;
        if (n_ranges == 0 && M >= 65536) {
            const int nx = xcd_splits_prepacked((long)B * ((N + 255) / 256), M, kFwdSlots, 32);
            nw = nw > nx ? nw : nx;
        }
        const size_t ws = (size_t)(nw < 2 ? 0 : nw) * (size_t)B * (size_t)N * (size_t)(D + 1) * sizeof(float) + 256 +
                          (size_t)B * (size_t)((M + 31) / 32) * 2048 + (size_t)4 * B * M * sizeof(float);
        bytes = bytes > ws ? bytes : ws;
    }
    if (n_ranges > 0) bytes += chunk_table_bytes(n_ranges, N, 64);   // row-chunk table of block-sparse launches (64-row tiles: the smallest, GLHIP_FLAG_SMALL_ROW_BLOCKS)
    if (autosort_applies(B, N, M, D, n_ranges, 0)) {
        // big dense distance reductions (p = 1, laplacian, energy) sort their clouds inside the workspace and run as a block-sparse launch
        // over slabs of 256 rows (glhip_autosort.h): the sorted copies + what that inner launch asks for
        const size_t sorted = autosort_bytes(N, M, D) + glhip_workspace_bytes(1, N, M, D, (N + kSortSlab - 1) / kSortSlab);
        bytes = bytes > sorted ? bytes : sorted;
    }
    return bytes;
}

const char* glhip_last_error(void) { return g_err; }

int glhip_softmin_fwd(const void* x, const void* y, const float* h, float* out, int B, int N, int M, int D,
                      float eps, int p, int in_dtype, const int32_t* ranges_i, const int32_t* slices_i,
                      const int32_t* redranges_j, int n_ranges, void* workspace, size_t workspace_bytes, int flags,
                      void* stream) {
    int rc = check_common("glhip_softmin_fwd", x, y, h, B, N, M, D, in_dtype, ranges_i, slices_i, redranges_j, n_ranges);
    if (rc) return rc;
    if (B == 0 || N == 0) return GLHIP_OK;   // nothing to write
    if (!out) return fail(GLHIP_EINVAL, "glhip_softmin_fwd: NULL out");
    if (!(eps > 0.f)) return fail(GLHIP_EINVAL, "glhip_softmin_fwd: eps must be > 0");
    if (p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "glhip_softmin_fwd: p must be 1 or 2 (got %d)", p);
    const Ranges rg{ranges_i, slices_i, redranges_j};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p == 1 && autosort_applies(B, N, M, D, n_ranges, flags)) {      // glhip_autosort.h: compact row blocks -> distances on the matrix cores
        AutoSort a;
        const int C = (N + kSortSlab - 1) / kSortSlab;
        rc = autosort_prepare(a, x, y, N, M, D, in_dtype, workspace, workspace_bytes, glhip_workspace_bytes(1, N, M, D, C), st);
        if (rc) return rc;
        if (a.on) {
            gather_f32(h, a.perm_y, a.col0, M, st);
            rc = glhip_softmin_fwd(a.xs, a.ys, a.col0, a.out, 1, N, M, D, eps, p, in_dtype, a.ranges_i, a.slices_i, a.red, a.C, a.inner_ws,
                                   a.inner_bytes, flags | GLHIP_FLAG_MFMA_DIST | GLHIP_FLAG_NO_SORT, stream);
            if (rc) return rc;
            scatter_f32(a.out, a.perm_x, out, N, st);
            return check_launch("glhip_softmin_fwd");
        }
    }
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags, n_ranges, N);
    rc = (in_dtype == GLHIP_F32)
             ? softmin_typed<false, float>(x, y, h, out, nullptr, nullptr, nullptr, B, N, M, D, eps, p, rg, n_ranges, sc, flags, st)
             : softmin_typed<false, bf16_t>(x, y, h, out, nullptr, nullptr, nullptr, B, N, M, D, eps, p, rg, n_ranges, sc, flags, st);
    return rc ? rc : check_launch("glhip_softmin_fwd");
}

int glhip_sinkhorn_step(const void* x, const void* y, const float* logw, const float* pot, const float* prev,
                        float* out, int B, int N, int M, int D, float eps, float damping, int p, int in_dtype,
                        const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges,
                        void* workspace, size_t workspace_bytes, int flags, void* stream) {
    int rc = check_common("glhip_sinkhorn_step", x, y, logw, B, N, M, D, in_dtype, ranges_i, slices_i, redranges_j, n_ranges);
    if (rc) return rc;
    if (B == 0 || N == 0) return GLHIP_OK;   // nothing to write
    if (!out) return fail(GLHIP_EINVAL, "glhip_sinkhorn_step: NULL out");
    if (out == prev) return fail(GLHIP_EINVAL, "glhip_sinkhorn_step: out must not alias prev (updates are simultaneous)");
    if (!(eps > 0.f)) return fail(GLHIP_EINVAL, "glhip_sinkhorn_step: eps must be > 0");
    if (p != 1 && p != 2) return fail(GLHIP_EUNSUPPORTED, "glhip_sinkhorn_step: p must be 1 or 2 (got %d)", p);
    const Ranges rg{ranges_i, slices_i, redranges_j};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p == 1 && autosort_applies(B, N, M, D, n_ranges, flags)) {      // as glhip_softmin_fwd; the potentials travel with their clouds
        AutoSort a;
        const int C = (N + kSortSlab - 1) / kSortSlab;
        rc = autosort_prepare(a, x, y, N, M, D, in_dtype, workspace, workspace_bytes, glhip_workspace_bytes(1, N, M, D, C), st);
        if (rc) return rc;
        if (a.on) {
            gather_f32(logw, a.perm_y, a.col0, M, st);
            if (pot) gather_f32(pot, a.perm_y, a.col1, M, st);
            if (prev) gather_f32(prev, a.perm_x, a.row0, N, st);
            rc = glhip_sinkhorn_step(a.xs, a.ys, a.col0, pot ? a.col1 : nullptr, prev ? a.row0 : nullptr, a.out, 1, N, M, D, eps, damping, p, in_dtype,
                                     a.ranges_i, a.slices_i, a.red, a.C, a.inner_ws, a.inner_bytes, flags | GLHIP_FLAG_MFMA_DIST | GLHIP_FLAG_NO_SORT, stream);
            if (rc) return rc;
            scatter_f32(a.out, a.perm_x, out, N, st);
            return check_launch("glhip_sinkhorn_step");
        }
    }
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags, n_ranges, N);
    StepArgs step;
    step.pot = pot;
    step.prev = prev;
    step.alpha = prev ? 0.5f * damping : damping;
    step.beta = prev ? 0.5f : 0.f;
    rc = (in_dtype == GLHIP_F32)
             ? softmin_typed<false, float>(x, y, logw, out, nullptr, nullptr, nullptr, B, N, M, D, eps, p, rg, n_ranges, sc, flags, st, step)
             : softmin_typed<false, bf16_t>(x, y, logw, out, nullptr, nullptr, nullptr, B, N, M, D, eps, p, rg, n_ranges, sc, flags, st, step);
    return rc ? rc : check_launch("glhip_sinkhorn_step");
}

int glhip_sinkhorn_iter4(const void* x, const void* y, const float* a_log, const float* b_log, const float* f_ba,
                         const float* g_ab, const float* f_aa, const float* g_bb, float* f_ba_out, float* g_ab_out,
                         float* f_aa_out, float* g_bb_out, int B, int N, int M, int D, float eps, float damping, int p,
                         int in_dtype, int first, void* workspace, size_t workspace_bytes, int flags, void* stream) {
    int rc = check_common("glhip_sinkhorn_iter4", x, y, b_log, B, N, M, D, in_dtype, nullptr, nullptr, nullptr, 0);
    if (rc) return rc;
    if ((p != 1 && p != 2) || D > kXdMaxD) return fail(GLHIP_EUNSUPPORTED, "glhip_sinkhorn_iter4: only p = 1, 2 and D <= 16 (got p = %d, D = %d)", p, D);
    if (flags & (GLHIP_FLAG_DIRECT | GLHIP_FLAG_NO_MFMA | GLHIP_FLAG_F32_MFMA | GLHIP_FLAG_XDL16))
        return fail(GLHIP_EUNSUPPORTED, "glhip_sinkhorn_iter4: runs on the default 32x32x16 kernel only (flags = %d)", flags);
    if (B == 0 || N == 0 || M == 0) return GLHIP_OK;
    if (!a_log || !f_ba_out || !g_ab_out) return fail(GLHIP_EINVAL, "glhip_sinkhorn_iter4: NULL a_log / f_ba_out / g_ab_out");
    if ((f_aa_out == nullptr) != (g_bb_out == nullptr)) return fail(GLHIP_EINVAL, "glhip_sinkhorn_iter4: f_aa_out and g_bb_out go together");
    if (first < 0 || first > 2) return fail(GLHIP_EINVAL, "glhip_sinkhorn_iter4: first must be 0, 1 or 2 (got %d)", first);
    if (first != 1 && (!f_ba || !g_ab || (f_aa_out && (!f_aa || !g_bb))))
        return fail(GLHIP_EINVAL, "glhip_sinkhorn_iter4: NULL potential (only allowed with first = 1)");
    if (first != 1 && (f_ba_out == f_ba || g_ab_out == g_ab || (f_aa_out && (f_aa_out == f_aa || g_bb_out == g_bb))))
        return fail(GLHIP_EINVAL, "glhip_sinkhorn_iter4: outputs must not alias inputs (updates are simultaneous)");
    if (!(eps > 0.f)) return fail(GLHIP_EINVAL, "glhip_sinkhorn_iter4: eps must be > 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags & ~GLHIP_FLAG_PREPACK, 0, N);
    rc = (in_dtype == GLHIP_F32)
             ? iter4_typed<float>(x, y, a_log, b_log, f_ba, g_ab, f_aa, g_bb, f_ba_out, g_ab_out, f_aa_out, g_bb_out, B, N, M, D, eps, damping, p, first, sc, st)
             : iter4_typed<bf16_t>(x, y, a_log, b_log, f_ba, g_ab, f_aa, g_bb, f_ba_out, g_ab_out, f_aa_out, g_bb_out, B, N, M, D, eps, damping, p, first, sc, st);
    return rc ? rc : check_launch("glhip_sinkhorn_iter4");
}

int glhip_sinkhorn_anneal(const void* x, const void* y, const float* a_log, const float* b_log, float* const* set0, float* const* set1,
                          int B, int N, int M, int D, const float* eps, const float* damping, int n_eps, int p, int in_dtype,
                          void* workspace, size_t workspace_bytes, int flags, float f16x2_min_eps, void* stream) {
    const char* fn = "glhip_sinkhorn_anneal";
    int rc = check_common(fn, x, y, b_log, B, N, M, D, in_dtype, nullptr, nullptr, nullptr, 0);
    if (rc) return rc;
    if ((p != 1 && p != 2) || D > kXdMaxD) return fail(GLHIP_EUNSUPPORTED, "%s: only p = 1, 2 and D <= 16 (got p = %d, D = %d)", fn, p, D);
    if (flags & (GLHIP_FLAG_DIRECT | GLHIP_FLAG_NO_MFMA | GLHIP_FLAG_F32_MFMA | GLHIP_FLAG_XDL16))
        return fail(GLHIP_EUNSUPPORTED, "%s: runs on the default 32x32x16 kernel only (flags = %d)", fn, flags);
    if (n_eps < 1 || !eps || !damping) return fail(GLHIP_EINVAL, "%s: needs n_eps >= 1 temperatures and dampings (n_eps = %d)", fn, n_eps);
    if (B == 0 || N == 0 || M == 0) return GLHIP_OK;
    if (!a_log || !set0 || !set1) return fail(GLHIP_EINVAL, "%s: NULL a_log / set0 / set1", fn);
    float* const* sets[2] = {set0, set1};
    const bool debias = set0[2] != nullptr;
    for (int s = 0; s < 2; ++s) {
        if (!sets[s][0] || !sets[s][1]) return fail(GLHIP_EINVAL, "%s: NULL f_ba / g_ab buffer in set%d", fn, s);
        if ((sets[s][2] != nullptr) != debias || (sets[s][3] != nullptr) != debias)
            return fail(GLHIP_EINVAL, "%s: the f_aa / g_bb buffers go together, in both sets or in neither", fn);
    }
    for (int i = 0; i < 8; ++i)
        for (int j = i + 1; j < 8; ++j)
            if (sets[i / 4][i % 4] && sets[i / 4][i % 4] == sets[j / 4][j % 4])
                return fail(GLHIP_EINVAL, "%s: a buffer appears twice in set0 / set1 (updates are simultaneous)", fn);
    for (int i = 0; i < n_eps; ++i)
        if (!(eps[i] > 0.f)) return fail(GLHIP_EINVAL, "%s: eps[%d] must be > 0", fn, i);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto one = [&](int i, int first, float* const* src, float* const* dst) {
        int fl = flags & ~GLHIP_FLAG_PREPACK;
        if (!(eps[i] >= f16x2_min_eps)) fl &= ~GLHIP_FLAG_F16X2;
        const Scratch sc = make_scratch(workspace, workspace_bytes, fl, 0, N);
        const float* s0 = src ? src[0] : nullptr; const float* s1 = src ? src[1] : nullptr;
        const float* s2 = src ? src[2] : nullptr; const float* s3 = src ? src[3] : nullptr;
        return (in_dtype == GLHIP_F32)
                   ? iter4_typed<float>(x, y, a_log, b_log, s0, s1, s2, s3, dst[0], dst[1], dst[2], dst[3], B, N, M, D, eps[i], damping[i], p, first, sc, st)
                   : iter4_typed<bf16_t>(x, y, a_log, b_log, s0, s1, s2, s3, dst[0], dst[1], dst[2], dst[3], B, N, M, D, eps[i], damping[i], p, first, sc, st);
    };
    rc = one(0, 1, nullptr, set0);      // initial potentials at the first temperature
    for (int i = 0; i < n_eps && !rc; ++i) rc = one(i, 0, sets[i % 2], sets[(i + 1) % 2]);
    return rc ? rc : check_launch(fn);
}

int glhip_sinkhorn_extrapolate4(const void* x, const void* y, const void* xc, const void* yc, const float* a_log_c, const float* b_log_c,
                                const float* f_ba, const float* g_ab, const float* f_aa, const float* g_bb, float* f_ba_out,
                                float* g_ab_out, float* f_aa_out, float* g_bb_out, int B, int N, int M, int Nc, int Mc, int D, float eps,
                                float damping, int p, int in_dtype, void* workspace, size_t workspace_bytes, int flags, void* stream) {
    const char* fn = "glhip_sinkhorn_extrapolate4";
    int rc = check_common(fn, x, yc, b_log_c, B, N, Mc, D, in_dtype, nullptr, nullptr, nullptr, 0);
    if (!rc) rc = check_common(fn, y, xc, a_log_c, B, M, Nc, D, in_dtype, nullptr, nullptr, nullptr, 0);
    if (rc) return rc;
    if ((p != 1 && p != 2) || D > kXdMaxD) return fail(GLHIP_EUNSUPPORTED, "%s: only p = 1, 2 and D <= 16 (got p = %d, D = %d)", fn, p, D);
    if (flags & (GLHIP_FLAG_DIRECT | GLHIP_FLAG_NO_MFMA | GLHIP_FLAG_F32_MFMA | GLHIP_FLAG_XDL16))
        return fail(GLHIP_EUNSUPPORTED, "%s: runs on the default 32x32x16 kernel only (flags = %d)", fn, flags);
    if (B == 0 || (N == 0 && M == 0)) return GLHIP_OK;
    if (Nc == 0 || Mc == 0) return fail(GLHIP_EINVAL, "%s: empty coarse measure (Nc = %d, Mc = %d)", fn, Nc, Mc);
    if (!f_ba_out || !g_ab_out || !f_ba || !g_ab) return fail(GLHIP_EINVAL, "%s: NULL f_ba / g_ab / f_ba_out / g_ab_out", fn);
    if ((f_aa_out == nullptr) != (g_bb_out == nullptr)) return fail(GLHIP_EINVAL, "%s: f_aa_out and g_bb_out go together", fn);
    if (f_aa_out && (!f_aa || !g_bb)) return fail(GLHIP_EINVAL, "%s: NULL f_aa / g_bb", fn);
    if (!(eps > 0.f)) return fail(GLHIP_EINVAL, "%s: eps must be > 0", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Scratch sc = make_scratch(workspace, workspace_bytes, flags & ~GLHIP_FLAG_PREPACK, 0, N > M ? N : M);
    rc = (in_dtype == GLHIP_F32)
             ? extrapolate4_typed<float>(x, y, xc, yc, a_log_c, b_log_c, f_ba, g_ab, f_aa, g_bb, f_ba_out, g_ab_out, f_aa_out, g_bb_out, B, N, M, Nc, Mc, D, eps, damping, p, sc, st)
             : extrapolate4_typed<bf16_t>(x, y, xc, yc, a_log_c, b_log_c, f_ba, g_ab, f_aa, g_bb, f_ba_out, g_ab_out, f_aa_out, g_bb_out, B, N, M, Nc, Mc, D, eps, damping, p, sc, st);
    return rc ? rc : check_launch(fn);
}

}  // extern "C"
