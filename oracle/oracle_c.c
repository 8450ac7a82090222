/*
 * oracle_c.c — plain-C (double precision, OpenMP over rows) restatement of the two reductions of
 * geomloss's hot path.  TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py through oracle/oracle_c.py; never by geomloss_amd/.
 *
 * Follows (paths under /root/reference/src/geomloss/_legacy/):
 *   softmin   : softmin_tensorized, sinkhorn_samples.py:70-71, on the cost of cost_routines :26-29
 *               (C = |x-y|^2/2 for p=2, |x-y| for p=1; utils.py:26-61), evaluated row by row with a
 *               two-pass max / sum-exp instead of a materialised matrix.
 *   kconv     : K @ v with the kernels of kernel_samples.py:62-82.
 *   *_grad_x  : what autograd returns for d/dx sum_i g_i out_i through those expressions.
 * Pinned against the reference's tensorized backend through tests/golden (see oracle_np.py).
 * Block-sparse variants take the KeOps-style ranges of include/glhip.h.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline double cost_p(const double* xi, const double* yj, int D, int p) {
    double d2 = 0.0;
    for (int d = 0; d < D; ++d) { const double t = xi[d] - yj[d]; d2 += t * t; }
    return p == 2 ? 0.5 * d2 : sqrt(d2 > 1e-8 ? d2 : 1e-8); /* clamp: utils.py:61 */
}

/* iterate the columns a row may see: dense [0,M) or the intervals of its block */
typedef struct { const int32_t* red; int q0, q1; int M; } colset;

static colset cols_of_row(int i, int M, const int32_t* ranges_i, const int32_t* slices_i,
                          const int32_t* redranges_j, int n_ranges, int* found) {
    colset c = {redranges_j, 0, 0, M};
    *found = 1;
    if (n_ranges == 0) { c.red = NULL; return c; }
    for (int k = 0; k < n_ranges; ++k)
        if (i >= ranges_i[2 * k] && i < ranges_i[2 * k + 1]) {
            c.q0 = k ? slices_i[k - 1] : 0;
            c.q1 = slices_i[k];
            return c;
        }
    *found = 0;
    return c;
}

#define FOR_COLS(c, j)                                                                         \
    for (int _q = (c).red ? (c).q0 : 0; _q < ((c).red ? (c).q1 : 1); ++_q)                      \
        for (int j = (c).red ? (c).red[2 * _q] : 0, _e = (c).red ? (c).red[2 * _q + 1] : (c).M; j < _e; ++j)

void oracle_softmin(const double* x, const double* y, const double* h, double* out, int N, int M, int D,
                    double eps, int p, const int32_t* ranges_i, const int32_t* slices_i,
                    const int32_t* redranges_j, int n_ranges) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < N; ++i) {
        int found;
        colset c = cols_of_row(i, M, ranges_i, slices_i, redranges_j, n_ranges, &found);
        if (!found) continue;
        const double* xi = x + (size_t)i * D;
        double m = -INFINITY;
        FOR_COLS(c, j) { const double v = h[j] - cost_p(xi, y + (size_t)j * D, D, p) / eps; if (v > m) m = v; }
        double s = 0.0;
        if (isfinite(m)) { FOR_COLS(c, j) s += exp(h[j] - cost_p(xi, y + (size_t)j * D, D, p) / eps - m); }
        out[i] = isfinite(m) ? -eps * (m + log(s)) : -eps * m;
    }
}

void oracle_softmin_grad_x(const double* x, const double* y, const double* h, const double* g, double* gx,
                           int N, int M, int D, double eps, int p, const int32_t* ranges_i,
                           const int32_t* slices_i, const int32_t* redranges_j, int n_ranges) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < N; ++i) {
        int found;
        colset c = cols_of_row(i, M, ranges_i, slices_i, redranges_j, n_ranges, &found);
        if (!found) continue;
        const double* xi = x + (size_t)i * D;
        double m = -INFINITY;
        FOR_COLS(c, j) { const double v = h[j] - cost_p(xi, y + (size_t)j * D, D, p) / eps; if (v > m) m = v; }
        double s = 0.0;
        double acc[64];
        for (int d = 0; d < D && d < 64; ++d) acc[d] = 0.0;
        FOR_COLS(c, j) {
            const double* yj = y + (size_t)j * D;
            const double w = exp(h[j] - cost_p(xi, yj, D, p) / eps - m);
            s += w;
            double inv = 1.0;
            if (p == 1) { const double d2 = 2.0 * cost_p(xi, yj, D, 2); inv = d2 > 1e-8 ? 1.0 / sqrt(d2) : 0.0; }
            for (int d = 0; d < D && d < 64; ++d) acc[d] += w * inv * (xi[d] - yj[d]);
        }
        for (int d = 0; d < D && d < 64; ++d) gx[(size_t)i * D + d] = s > 0.0 ? g[i] * acc[d] / s : 0.0;
    }
}

static inline double kern(int kind, double d2, double blur) {
    if (kind == 0) return exp(-d2 / (2.0 * blur * blur));
    if (kind == 1) { const double s2 = d2 / (blur * blur); return exp(-sqrt(s2 > 1e-8 ? s2 : 1e-8)); } /* utils.py:61 on x/blur */
    return -sqrt(d2 > 1e-8 ? d2 : 1e-8);
}

void oracle_kconv(int kind, const double* x, const double* y, const double* v, double* out, int N, int M, int D,
                  double blur, const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j,
                  int n_ranges) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < N; ++i) {
        int found;
        colset c = cols_of_row(i, M, ranges_i, slices_i, redranges_j, n_ranges, &found);
        if (!found) continue;
        const double* xi = x + (size_t)i * D;
        double acc = 0.0;
        FOR_COLS(c, j) acc += kern(kind, 2.0 * cost_p(xi, y + (size_t)j * D, D, 2), blur) * v[j];
        out[i] = acc;
    }
}

void oracle_kconv_grad_x(int kind, const double* x, const double* y, const double* v, const double* g, double* gx,
                         int N, int M, int D, double blur, const int32_t* ranges_i, const int32_t* slices_i,
                         const int32_t* redranges_j, int n_ranges) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < N; ++i) {
        int found;
        colset c = cols_of_row(i, M, ranges_i, slices_i, redranges_j, n_ranges, &found);
        if (!found) continue;
        const double* xi = x + (size_t)i * D;
        double acc[64];
        for (int d = 0; d < D && d < 64; ++d) acc[d] = 0.0;
        FOR_COLS(c, j) {
            const double* yj = y + (size_t)j * D;
            const double d2 = 2.0 * cost_p(xi, yj, D, 2);
            const double dist = sqrt(d2);
            double coef; /* dk/dx = coef * (x - y) */
            if (kind == 0) coef = -exp(-d2 / (2.0 * blur * blur)) / (blur * blur);
            else if (kind == 1) coef = d2 > 1e-8 * blur * blur ? -exp(-dist / blur) / (blur * dist) : 0.0;
            else coef = d2 > 1e-8 ? -1.0 / dist : 0.0;
            for (int d = 0; d < D && d < 64; ++d) acc[d] += v[j] * coef * (xi[d] - yj[d]);
        }
        for (int d = 0; d < D && d < 64; ++d) gx[(size_t)i * D + d] = g[i] * acc[d];
    }
}
