/*
 * tests/cabi/smoke.c — a caller of libgeomloss_hip.so that is NOT geomloss_amd/hip.py: plain C, the prototypes of
 * include/glhip.h as the compiler sees them, device memory from hipMalloc, no Python, no torch.
 *
 * Every other test reaches the library through the ctypes signature table of hip.py; a wrong `argtypes` entry there and a wrong
 * prototype here cannot agree with each other by accident, so this program pins the header itself (round-4 review, weak #9).
 * It plays the reference's call sites for the path: `lse_genred(...)(x, y, h, 1/eps)` (_legacy/sinkhorn_samples.py:322-346),
 * `keops_lse(..., ranges=...)` (:432-450), KeOps `Grad` of the last soft-min, and `K @ v` (kernel_samples.py:117-137).
 *
 * Checker: the C restatement of the oracle (oracle/oracle_c.c, linked as liboracle_c.so) on the same doubles-of-floats.
 * TEST INFRASTRUCTURE: built by __graft_entry__.build() (tests/cabi/Makefile), run by tests/test_cabi_gpu.py on the GPU box.
 * Prints one line per check and exits non-zero on the first failure.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "glhip.h"

/* oracle/oracle_c.c */
void oracle_softmin(const double* x, const double* y, const double* h, double* out, int N, int M, int D, double eps, int p,
                    const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges);
void oracle_softmin_grad_x(const double* x, const double* y, const double* h, const double* g, double* gx, int N, int M, int D,
                           double eps, int p, const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j,
                           int n_ranges);
void oracle_kconv(int kind, const double* x, const double* y, const double* v, double* out, int N, int M, int D, double blur,
                  const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges);

#define HIP_OK(call)                                                                                 \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
            exit(2);                                                                                 \
        }                                                                                            \
    } while (0)
#define GL_OK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != GLHIP_OK) {                                                                       \
            fprintf(stderr, "glhip error %d (%s) at %s:%d\n", rc_, glhip_last_error(), __FILE__, __LINE__); \
            exit(3);                                                                                 \
        }                                                                                            \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double uniform(void) { /* xorshift64*: deterministic, no libc rand */
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}
static double normal(void) { return sqrt(-2.0 * log(1.0 - uniform())) * cos(6.283185307179586 * uniform()); }

static void* to_device(const void* src, size_t bytes) {
    void* p = NULL;
    HIP_OK(hipMalloc(&p, bytes ? bytes : 4));
    if (bytes && src) HIP_OK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));      /* src == NULL: an uninitialised buffer */
    return p;
}

static int failures = 0;
static void report(const char* what, double err, double tol) {
    printf("%-58s err %.3e  (tol %.1e)  %s\n", what, err, tol, err < tol ? "ok" : "FAIL");
    if (!(err < tol)) ++failures;
}
static double max_abs(const float* got, const double* want, size_t n, int relative) {
    double e = 0.0, s = 0.0;
    for (size_t k = 0; k < n; ++k) {
        if (isinf(want[k]) && isinf((double)got[k]) && (want[k] > 0) == (got[k] > 0)) continue;
        const double d = fabs((double)got[k] - want[k]);
        if (!(d <= e)) e = d; /* NaN counts */
        if (fabs(want[k]) > s && !isinf(want[k])) s = fabs(want[k]);
    }
    return relative ? e / (s > 0 ? s : 1.0) : e;
}

int main(void) {
    enum { N = 700, M = 900, D = 3 };
    const float eps = 0.05f * 0.05f, blur = 0.2f;
    int n_dev = 0;
    HIP_OK(hipGetDeviceCount(&n_dev));
    if (n_dev < 1) { fprintf(stderr, "no GPU\n"); return 2; }
    HIP_OK(hipSetDevice(0));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    printf("glhip_version() = %d (header GLHIP_VERSION = %d)\n", glhip_version(), GLHIP_VERSION);
    if (glhip_version() != GLHIP_VERSION) { fprintf(stderr, "header / library version mismatch\n"); return 4; }

    /* inputs: floats on the device, the same values as doubles for the oracle */
    static float xf[N * D], yf[M * D], hf[M], gf[N], vf[M];
    static double xd[N * D], yd[M * D], hd[M], gd[N], vd[M];
    for (int k = 0; k < N * D; ++k) xd[k] = xf[k] = (float)uniform();
    for (int k = 0; k < M * D; ++k) yd[k] = yf[k] = (float)(0.1 + 0.8 * uniform());
    for (int k = 0; k < M; ++k) hd[k] = hf[k] = (float)(normal() - log((double)M));
    for (int k = 0; k < N; ++k) gd[k] = gf[k] = (float)normal();
    for (int k = 0; k < M; ++k) vd[k] = vf[k] = (float)(uniform() / M) * ((k % 7) ? 1.f : -1.f);
    hd[11] = hf[11] = -INFINITY; /* a column without mass */
    void *x = to_device(xf, sizeof xf), *y = to_device(yf, sizeof yf);
    float *h = to_device(hf, sizeof hf), *g = to_device(gf, sizeof gf), *v = to_device(vf, sizeof vf);
    float *out = to_device(NULL, N * sizeof(float)), *gx = to_device(NULL, N * D * sizeof(float));
    static float out_h[N], gx_h[N * D];
    static double ref[N], refg[N * D];

    const size_t ws_bytes = glhip_workspace_bytes(1, N, M, D, 0);
    void* ws = to_device(NULL, ws_bytes);

    /* 1. dense soft-min, p = 2 and p = 1 (lse_genred, sinkhorn_samples.py:322-346) */
    for (int p = 2; p >= 1; --p) {
        const float e = p == 2 ? eps : 0.05f;
        GL_OK(glhip_softmin_fwd(x, y, h, out, 1, N, M, D, e, p, GLHIP_F32, NULL, NULL, NULL, 0, ws, ws_bytes, 0, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipMemcpy(out_h, out, sizeof out_h, hipMemcpyDeviceToHost));
        oracle_softmin(xd, yd, hd, ref, N, M, D, (double)e, p, NULL, NULL, NULL, 0);
        report(p == 2 ? "glhip_softmin_fwd dense p=2" : "glhip_softmin_fwd dense p=1", max_abs(out_h, ref, N, 0), 3e-6);
    }

    /* 2. its row gradient (KeOps Grad of the last soft-min, sinkhorn_divergence.py:612-623) */
    GL_OK(glhip_softmin_fwd(x, y, h, out, 1, N, M, D, eps, 2, GLHIP_F32, NULL, NULL, NULL, 0, ws, ws_bytes, 0, stream));
    GL_OK(glhip_softmin_bwd_x(x, y, h, out, g, gx, 1, N, M, D, eps, 2, GLHIP_F32, NULL, NULL, NULL, 0, ws, ws_bytes, 0, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(gx_h, gx, sizeof gx_h, hipMemcpyDeviceToHost));
    oracle_softmin_grad_x(xd, yd, hd, gd, refg, N, M, D, (double)eps, 2, NULL, NULL, NULL, 0);
    report("glhip_softmin_bwd_x dense p=2 (relative, max-norm)", max_abs(gx_h, refg, N * D, 1), 2e-5);

    /* 3. block-sparse soft-min (keops_lse with ranges, sinkhorn_samples.py:432-450): 3 row blocks; the second keeps nothing */
    {
        const int32_t ranges_i[6] = {0, 250, 250, 300, 300, N}, slices_i[3] = {2, 2, 3}, red[6] = {0, 100, 400, 650, 500, M};
        int32_t *ri = to_device(ranges_i, sizeof ranges_i), *si = to_device(slices_i, sizeof slices_i), *rj = to_device(red, sizeof red);
        const size_t wsb = glhip_workspace_bytes(1, N, M, D, 3);
        void* wsr = to_device(NULL, wsb);
        GL_OK(glhip_softmin_fwd(x, y, h, out, 1, N, M, D, eps, 2, GLHIP_F32, ri, si, rj, 3, wsr, wsb, 0, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipMemcpy(out_h, out, sizeof out_h, hipMemcpyDeviceToHost));
        oracle_softmin(xd, yd, hd, ref, N, M, D, (double)eps, 2, ranges_i, slices_i, red, 3);
        report("glhip_softmin_fwd block-sparse (one empty row block)", max_abs(out_h, ref, N, 0), 3e-6);
        int inf_rows = 0;
        for (int i = 250; i < 300; ++i) inf_rows += isinf(out_h[i]) && out_h[i] > 0;
        report("   rows of the empty block are +inf (missing of 50)", (double)(50 - inf_rows), 0.5);
        HIP_OK(hipFree(ri)); HIP_OK(hipFree(si)); HIP_OK(hipFree(rj)); HIP_OK(hipFree(wsr));
    }

    /* 4. kernel products K @ v (kernel_samples.py:117-137) */
    for (int kind = GLHIP_GAUSSIAN; kind <= GLHIP_ENERGY; ++kind) {
        GL_OK(glhip_kernel_conv_fwd(kind, x, y, v, out, 1, N, M, D, blur, GLHIP_F32, NULL, NULL, NULL, 0, ws, ws_bytes, 0, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipMemcpy(out_h, out, sizeof out_h, hipMemcpyDeviceToHost));
        oracle_kconv(kind, xd, yd, vd, ref, N, M, D, (double)blur, NULL, NULL, NULL, 0);
        char name[64];
        snprintf(name, sizeof name, "glhip_kernel_conv_fwd kind=%d (relative, max-norm)", kind);
        report(name, max_abs(out_h, ref, N, 1), 1e-4);
    }

    /* 5. no workspace at all (NULL, 0) is legal: same answer */
    GL_OK(glhip_softmin_fwd(x, y, h, out, 1, N, M, D, eps, 2, GLHIP_F32, NULL, NULL, NULL, 0, NULL, 0, 0, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(out_h, out, sizeof out_h, hipMemcpyDeviceToHost));
    oracle_softmin(xd, yd, hd, ref, N, M, D, (double)eps, 2, NULL, NULL, NULL, 0);
    report("glhip_softmin_fwd without workspace", max_abs(out_h, ref, N, 0), 3e-6);

    /* 5b. big dense DISTANCE reductions sort their clouds behind the ABI (glhip_autosort.h; N >= 65536, N M >= 5e8): p = 1 soft-min,
     *     laplacian and energy products at 70 000 x 66 000 — against the oracle on a sample of rows, and against the same call
     *     under GLHIP_FLAG_NO_SORT (the generic kernel on the caller's order): one answer, two kernels */
    {
        enum { NB = 70000, MB = 66000, NS = 96 };
        float* xb = malloc(sizeof(float) * NB * D), *yb = malloc(sizeof(float) * MB * D), *hb = malloc(sizeof(float) * MB), *vb = malloc(sizeof(float) * MB);
        float *o1 = malloc(sizeof(float) * NB), *o2 = malloc(sizeof(float) * NB);
        double *yq = malloc(sizeof(double) * MB * D), *hq = malloc(sizeof(double) * MB), *vq = malloc(sizeof(double) * MB);
        for (int k = 0; k < NB * D; ++k) xb[k] = (float)uniform();
        for (int k = 0; k < MB * D; ++k) yq[k] = yb[k] = (float)(0.1 + 0.8 * uniform());
        for (int k = 0; k < MB; ++k) { hq[k] = hb[k] = (float)(0.3 * normal() - log((double)MB)); vq[k] = vb[k] = (float)(uniform() / MB); }
        void *X = to_device(xb, sizeof(float) * NB * D), *Y = to_device(yb, sizeof(float) * MB * D);
        float *H = to_device(hb, sizeof(float) * MB), *V = to_device(vb, sizeof(float) * MB), *O = to_device(NULL, sizeof(float) * NB);
        const size_t wb = glhip_workspace_bytes(1, NB, MB, D, 0);
        void* W = to_device(NULL, wb);
        static double xs[NS * D], rs[NS];
        static float got[NS];
        for (int mode = 0; mode < 3; ++mode) {        /* 0: soft-min p = 1, 1: laplacian, 2: energy */
            for (int pass = 0; pass < 2; ++pass) {
                const int fl = pass ? GLHIP_FLAG_NO_SORT : 0;
                if (mode == 0) GL_OK(glhip_softmin_fwd(X, Y, H, O, 1, NB, MB, D, 0.05f, 1, GLHIP_F32, NULL, NULL, NULL, 0, W, wb, fl, stream));
                else GL_OK(glhip_kernel_conv_fwd(mode == 1 ? GLHIP_LAPLACIAN : GLHIP_ENERGY, X, Y, V, O, 1, NB, MB, D, 0.1f, GLHIP_F32, NULL, NULL, NULL, 0, W, wb, fl, stream));
                HIP_OK(hipStreamSynchronize(stream));
                HIP_OK(hipMemcpy(pass ? o2 : o1, O, sizeof(float) * NB, hipMemcpyDeviceToHost));
            }
            double between = 0.0, scale = 0.0;
            for (int i = 0; i < NB; ++i) { between = fmax(between, fabs((double)o1[i] - o2[i])); scale = fmax(scale, fabs((double)o2[i])); }
            for (int k = 0; k < NS; ++k) {
                const int i = (int)((long)k * (NB - 1) / (NS - 1));
                for (int d = 0; d < D; ++d) xs[k * D + d] = xb[i * D + d];
                got[k] = o1[i];
            }
            if (mode == 0) oracle_softmin(xs, yq, hq, rs, NS, MB, D, 0.05, 1, NULL, NULL, NULL, 0);
            else oracle_kconv(mode == 1 ? GLHIP_LAPLACIAN : GLHIP_ENERGY, xs, yq, vq, rs, NS, MB, D, 0.1, NULL, NULL, NULL, 0);
            char name[96];
            snprintf(name, sizeof name, "self-sorting dense launch, mode %d: sampled rows vs oracle (relative)", mode);
            report(name, max_abs(got, rs, NS, 1), 2e-5);
            snprintf(name, sizeof name, "   ... vs the same call under GLHIP_FLAG_NO_SORT (relative)");
            report(name, between / scale, 2e-5);
        }
        HIP_OK(hipFree(X)); HIP_OK(hipFree(Y)); HIP_OK(hipFree(H)); HIP_OK(hipFree(V)); HIP_OK(hipFree(O)); HIP_OK(hipFree(W));
        free(xb); free(yb); free(hb); free(vb); free(o1); free(o2); free(yq); free(hq); free(vq);
    }

    /* 5c. the fused half-step in double precision (glhip_sinkhorn_step_f64, round 6): out = (prev + damping * softmin(logw + pot / eps)) / 2 */
    {
        static double potd[M], prevd[N], hh[M], want[N], got64[N];
        for (int k = 0; k < M; ++k) { potd[k] = 0.05 * normal(); hh[k] = hd[k] + potd[k] / (double)eps; }
        for (int k = 0; k < N; ++k) prevd[k] = normal();
        double *X = to_device(xd, sizeof xd), *Y = to_device(yd, sizeof yd), *H = to_device(hd, sizeof hd);
        double *P = to_device(potd, sizeof potd), *V = to_device(prevd, sizeof prevd), *O = to_device(NULL, sizeof got64);
        GL_OK(glhip_sinkhorn_step_f64(X, Y, H, P, V, O, 1, N, M, D, (double)eps, 0.8, 2, NULL, NULL, NULL, 0, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipMemcpy(got64, O, sizeof got64, hipMemcpyDeviceToHost));
        oracle_softmin(xd, yd, hh, want, N, M, D, (double)eps, 2, NULL, NULL, NULL, 0);
        double worst = 0.0;
        for (int i = 0; i < N; ++i) {
            const double w = 0.5 * (prevd[i] + 0.8 * want[i]);
            if (fabs(got64[i] - w) > worst) worst = fabs(got64[i] - w);
        }
        report("glhip_sinkhorn_step_f64 (double precision half-step)", worst, 1e-12);
        HIP_OK(hipFree(X)); HIP_OK(hipFree(Y)); HIP_OK(hipFree(H)); HIP_OK(hipFree(P)); HIP_OK(hipFree(V)); HIP_OK(hipFree(O));
    }

    /* 5d. block-sparse ranges from a keep rule on cluster pairs (glhip_block_ranges, sinkhorn_samples.py:512-530): a symmetric pattern
     *     (rows = cols, f = g) built in one orientation only (slices_cols = red_rows = NULL, version 117) = the two-orientation call */
    {
        enum { C = 64 };
        static float cf[C * D], ff[C];
        static int32_t rr[C * 2];
        for (int k = 0; k < C * D; ++k) cf[k] = (float)uniform();
        for (int k = 0; k < C; ++k) { ff[k] = (float)(0.02 * normal()); rr[2 * k] = 10 * k; rr[2 * k + 1] = 10 * k + 7 + k % 3; }
        float *cc = to_device(cf, sizeof cf), *fd = to_device(ff, sizeof ff);
        int32_t* rd = to_device(rr, sizeof rr);
        const long long cap = (long long)C * ((C + 1) / 2);
        int32_t *s1 = to_device(NULL, C * 4), *r1 = to_device(NULL, (size_t)cap * 8), *s2 = to_device(NULL, C * 4), *r2 = to_device(NULL, (size_t)cap * 8);
        int32_t *s3 = to_device(NULL, C * 4), *r3 = to_device(NULL, (size_t)cap * 8), *st = to_device(NULL, 4);
        GL_OK(glhip_block_ranges(GLHIP_KEEP_DUAL_SLACK, cc, cc, fd, fd, C, C, D, 2, 0.05f, rd, rd, s1, r1, s2, r2, cap, st, stream));
        GL_OK(glhip_block_ranges(GLHIP_KEEP_DUAL_SLACK, cc, cc, fd, fd, C, C, D, 2, 0.05f, rd, rd, s3, r3, NULL, NULL, cap, st, stream));
        HIP_OK(hipStreamSynchronize(stream));
        static int32_t a1[C], a2[C], a3[C];
        HIP_OK(hipMemcpy(a1, s1, sizeof a1, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(a2, s2, sizeof a2, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(a3, s3, sizeof a3, hipMemcpyDeviceToHost));
        int bad = a1[C - 1] <= 0;      /* something is kept (the diagonal at least) */
        for (int k = 0; k < C; ++k) bad |= (a1[k] != a2[k]) | (a1[k] != a3[k]);
        const size_t nint = (size_t)a1[C - 1];
        int32_t *b1 = malloc(nint * 8), *b2 = malloc(nint * 8), *b3 = malloc(nint * 8);
        HIP_OK(hipMemcpy(b1, r1, nint * 8, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(b2, r2, nint * 8, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(b3, r3, nint * 8, hipMemcpyDeviceToHost));
        bad |= memcmp(b1, b2, nint * 8) != 0 || memcmp(b1, b3, nint * 8) != 0;
        printf("glhip_block_ranges: %zu intervals for %d x %d clusters\n", nint, C, C);
        report("glhip_block_ranges: symmetric pattern, one orientation = both orientations", bad ? 1.0 : 0.0, 0.5);
        free(b1); free(b2); free(b3);
        HIP_OK(hipFree(cc)); HIP_OK(hipFree(fd)); HIP_OK(hipFree(rd)); HIP_OK(hipFree(s1)); HIP_OK(hipFree(r1)); HIP_OK(hipFree(s2));
        HIP_OK(hipFree(r2)); HIP_OK(hipFree(s3)); HIP_OK(hipFree(r3)); HIP_OK(hipFree(st));
    }

    /* 6. error paths: code + thread-local message, nothing thrown across the ABI */
    {
        int rc = glhip_softmin_fwd(x, y, h, out, 1, N, M, D, eps, 3, GLHIP_F32, NULL, NULL, NULL, 0, ws, ws_bytes, 0, stream);
        printf("p = 3            -> rc %d, \"%s\"\n", rc, glhip_last_error());
        report("unsupported exponent returns GLHIP_EUNSUPPORTED", rc == GLHIP_EUNSUPPORTED ? 0.0 : 1.0, 0.5);
        rc = glhip_softmin_fwd(NULL, y, h, out, 1, N, M, D, eps, 2, GLHIP_F32, NULL, NULL, NULL, 0, ws, ws_bytes, 0, stream);
        printf("x = NULL         -> rc %d, \"%s\"\n", rc, glhip_last_error());
        report("NULL cloud returns GLHIP_EINVAL with a message", (rc == GLHIP_EINVAL && strlen(glhip_last_error()) > 0) ? 0.0 : 1.0, 0.5);
        rc = glhip_softmin_fwd(x, y, h, out, 1, N, M, D, -1.f, 2, GLHIP_F32, NULL, NULL, NULL, 0, ws, ws_bytes, 0, stream);
        report("eps <= 0 returns GLHIP_EINVAL", rc == GLHIP_EINVAL ? 0.0 : 1.0, 0.5);
    }

    HIP_OK(hipStreamDestroy(stream));
    printf(failures ? "C-ABI smoke: %d FAILED\n" : "C-ABI smoke: all checks passed\n", failures);
    return failures ? 1 : 0;
}
