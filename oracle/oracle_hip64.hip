// oracle_hip64.hip — brute-force FLOAT64 restatement of the reductions of geomloss's hot path, one GPU thread per row.
//
// TEST INFRASTRUCTURE ONLY (same rule as oracle_np.py / oracle_c.c): loaded by oracle/oracle_hip64.py for tests/, never by
// geomloss_amd/.  It exists because the BASELINE configs are 1e6 x 1e6 points: the chunked float64 torch oracle
// (oracle_torch64.py) needs 100-300 s per two-scale loss there, this one a few seconds.  It shares nothing with the kernels under
// test (csrc/): no matrix cores, no bf16 splitting, no expanded squared distance, no lazy maximum, no LDS tiling — every pair is
// evaluated on explicit float64 differences, exactly as the formulas of the reference read:
//   cost        C = |x-y|^2 / 2  (p = 2)  or  sqrt(max(|x-y|^2, 1e-8))  (p = 1)          _legacy/sinkhorn_samples.py:26-29, utils.py:26-61
//   soft-min    f_i = -eps log sum_j exp(h_j - C_ij / eps)                                 _legacy/sinkhorn_samples.py:32-71
//   its x-gradient  g_i sum_j P_ij dC/dx_i,  P = softmax_j(h_j - C_ij / eps)               (autograd through :70-71)
//   kernels     exp(-|x-y|^2 / 2 b^2) | exp(-sqrt(max(|x/b-y/b|^2, 1e-8))) | -sqrt(max(|x-y|^2, 1e-8))   _legacy/kernel_samples.py:62-82
//   products    (K v)_i and d/dx_i sum_i g_i (K v)_i                                       _legacy/kernel_samples.py:92-146
// Block-sparse reductions: row i reduces over the column intervals offsets[lab[i]] .. offsets[lab[i] + 1] of `intervals`
// (the kept blocks of kernel_truncation, sinkhorn_samples.py:493-530); lab == NULL means all columns.
// Pinned on the GPU box, before anything relies on it, against oracle_c.c and oracle_torch64.py (tests/test_full_size_gpu.py).
#include <hip/hip_runtime.h>
#include <math.h>

namespace {

constexpr int kMaxD = 16;
constexpr int kThreads = 128;

struct Cols {               // the columns row i reduces over
    const int* lab;         // (N) row -> pattern row, or NULL (dense)
    const int* offsets;     // (C + 1) CSR offsets into intervals
    const int* intervals;   // (nnz, 2) [start, end)
    int M;
};

__device__ inline int n_intervals(const Cols& c, int i, int& first) {
    if (!c.lab) { first = 0; return 1; }
    const int k = c.lab[i];
    first = c.offsets[k];
    return c.offsets[k + 1] - first;
}
__device__ inline void interval(const Cols& c, int q, int& j0, int& j1) {
    if (!c.lab) { j0 = 0; j1 = c.M; return; }
    j0 = c.intervals[2 * q];
    j1 = c.intervals[2 * q + 1];
}

// DD > 0: the dimension is a compile-time constant (1, 2, 3: register arrays, unrolled loops); DD = 0: any D <= kMaxD at run time
template <int DD>
__device__ inline double sqdist(const double* xi, const double* __restrict__ yj, int D) {
    double d2 = 0.0;
#pragma unroll
    for (int d = 0; d < (DD ? DD : D); ++d) {
        const double t = xi[d] - yj[d];
        d2 += t * t;
    }
    return d2;
}
__device__ inline double cost(double d2, int p) { return p == 2 ? 0.5 * d2 : sqrt(fmax(d2, 1e-8)); }

// log-sum-exp of u_ij = h_j - C_ij / eps over the row's columns: returns (max, sum of exp(u - max)); sum = 0 for no column.
// Four columns at a time: one rescaling exponential per four.
template <int DD>
__device__ inline void row_lse(const double* xi, const double* __restrict__ y, const double* __restrict__ h, int D, double eps, int p,
                               const Cols& c, int i, double& m, double& s) {
    m = -INFINITY;
    s = 0.0;
    int first;
    const int nq = n_intervals(c, i, first);
    for (int q = first; q < first + nq; ++q) {
        int j0, j1;
        interval(c, q, j0, j1);
        for (int j = j0; j < j1; j += 4) {
            double u[4], cm = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u[k] = -INFINITY;
                if (j + k < j1) u[k] = h[j + k] - cost(sqdist<DD>(xi, y + (long)(j + k) * D, D), p) / eps;
                cm = fmax(cm, u[k]);
            }
            if (!(cm > -INFINITY)) continue;          // four columns without mass (h = -inf)
            const double mn = fmax(m, cm);
            s *= exp(m - mn);                         // exp(-inf) = 0 the first time
#pragma unroll
            for (int k = 0; k < 4; ++k) s += exp(u[k] - mn);
            m = mn;
        }
    }
}

template <int DD>
__global__ void softmin_kernel(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ h, double* __restrict__ out,
                               int N, int D, double eps, int p, Cols c) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= N) return;
    double xi[DD ? DD : kMaxD];
#pragma unroll
    for (int d = 0; d < (DD ? DD : D); ++d) xi[d] = x[(long)i * D + d];
    double m, s;
    row_lse<DD>(xi, y, h, D, eps, p, c, i, m, s);
    out[i] = (s > 0.0) ? -eps * (m + log(s)) : INFINITY;
}

template <int DD>
__global__ void softmin_grad_kernel(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ h,
                                    const double* __restrict__ g, double* __restrict__ out, int N, int D, double eps, int p, Cols c) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= N) return;
    double xi[DD ? DD : kMaxD], acc[DD ? DD : kMaxD];
#pragma unroll
    for (int d = 0; d < (DD ? DD : D); ++d) { xi[d] = x[(long)i * D + d]; acc[d] = 0.0; }
    double m, s;
    row_lse<DD>(xi, y, h, D, eps, p, c, i, m, s);
    if (s > 0.0) {
        const double lse = m + log(s);
        int first;
        const int nq = n_intervals(c, i, first);
        for (int q = first; q < first + nq; ++q) {
            int j0, j1;
            interval(c, q, j0, j1);
            for (int j = j0; j < j1; ++j) {
                const double* yj = y + (long)j * D;
                const double d2 = sqdist<DD>(xi, yj, D);
                const double P = exp(h[j] - cost(d2, p) / eps - lse);
                // dC/dx_i = x_i - y_j (p = 2);  (x_i - y_j) / |x_i - y_j|, zero inside the clamp of utils.py:61 (p = 1)
                const double w = (p == 2) ? P : (d2 > 1e-8 ? P / sqrt(d2) : 0.0);
#pragma unroll
                for (int d = 0; d < (DD ? DD : D); ++d) acc[d] += w * (xi[d] - yj[d]);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < (DD ? DD : D); ++d) out[(long)i * D + d] = g[i] * acc[d];
}

// kernel value k(x_i, y_j) and the factor c_ij of its gradient dk/dx_i = c_ij (x_i - y_j);  kind: 0 gaussian, 1 laplacian, 2 energy
__device__ inline void kernel_pair(int kind, double d2, double blur, double& k, double& cf) {
    if (kind == 0) {
        k = exp(-d2 / (2.0 * blur * blur));
        cf = -k / (blur * blur);
    } else if (kind == 1) {
        const double s2 = d2 / (blur * blur);
        const double dist = sqrt(fmax(s2, 1e-8));
        k = exp(-dist);
        cf = (s2 > 1e-8) ? -k / (dist * blur * blur) : 0.0;
    } else {
        const double dist = sqrt(fmax(d2, 1e-8));
        k = -dist;
        cf = (d2 > 1e-8) ? -1.0 / dist : 0.0;
    }
}

template <int DD>
__global__ void kconv_kernel(int kind, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ v,
                             const double* __restrict__ g, double* __restrict__ out, double* __restrict__ gout, int N, int D, double blur,
                             Cols c) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= N) return;
    double xi[DD ? DD : kMaxD], acc[DD ? DD : kMaxD], sum = 0.0;
#pragma unroll
    for (int d = 0; d < (DD ? DD : D); ++d) { xi[d] = x[(long)i * D + d]; acc[d] = 0.0; }
    int first;
    const int nq = n_intervals(c, i, first);
    for (int q = first; q < first + nq; ++q) {
        int j0, j1;
        interval(c, q, j0, j1);
        for (int j = j0; j < j1; ++j) {
            const double* yj = y + (long)j * D;
            double k, cf;
            kernel_pair(kind, sqdist<DD>(xi, yj, D), blur, k, cf);
            sum += k * v[j];
            if (gout) {
                const double w = cf * v[j];
#pragma unroll
                for (int d = 0; d < (DD ? DD : D); ++d) acc[d] += w * (xi[d] - yj[d]);
            }
        }
    }
    if (out) out[i] = sum;
    if (gout) {
#pragma unroll
        for (int d = 0; d < (DD ? DD : D); ++d) gout[(long)i * D + d] = g[i] * acc[d];
    }
}

int check(int N, int M, int D) { return (N < 0 || M < 0 || D < 1 || D > kMaxD) ? -1 : 0; }

}  // namespace

extern "C" {

// all pointers: device memory, float64 (int32 for lab / offsets / intervals); synchronous on return; 0 = ok
int o64_softmin(const double* x, const double* y, const double* h, double* out, int N, int M, int D, double eps, int p,
                const int* lab, const int* offsets, const int* intervals) {
    if (check(N, M, D) || (p != 1 && p != 2) || !(eps > 0.0)) return -1;
    if (N == 0) return 0;
    const Cols c{lab, offsets, intervals, M};
    const dim3 grid((N + kThreads - 1) / kThreads), block(kThreads);
    if (D == 1) hipLaunchKernelGGL(softmin_kernel<1>, grid, block, 0, 0, x, y, h, out, N, D, eps, p, c);
    else if (D == 2) hipLaunchKernelGGL(softmin_kernel<2>, grid, block, 0, 0, x, y, h, out, N, D, eps, p, c);
    else if (D == 3) hipLaunchKernelGGL(softmin_kernel<3>, grid, block, 0, 0, x, y, h, out, N, D, eps, p, c);
    else hipLaunchKernelGGL(softmin_kernel<0>, grid, block, 0, 0, x, y, h, out, N, D, eps, p, c);
    return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? 0 : -2;
}

int o64_softmin_grad_x(const double* x, const double* y, const double* h, const double* g, double* out, int N, int M, int D, double eps,
                       int p, const int* lab, const int* offsets, const int* intervals) {
    if (check(N, M, D) || (p != 1 && p != 2) || !(eps > 0.0)) return -1;
    if (N == 0) return 0;
    const Cols c{lab, offsets, intervals, M};
    const dim3 grid((N + kThreads - 1) / kThreads), block(kThreads);
    if (D == 1) hipLaunchKernelGGL(softmin_grad_kernel<1>, grid, block, 0, 0, x, y, h, g, out, N, D, eps, p, c);
    else if (D == 2) hipLaunchKernelGGL(softmin_grad_kernel<2>, grid, block, 0, 0, x, y, h, g, out, N, D, eps, p, c);
    else if (D == 3) hipLaunchKernelGGL(softmin_grad_kernel<3>, grid, block, 0, 0, x, y, h, g, out, N, D, eps, p, c);
    else hipLaunchKernelGGL(softmin_grad_kernel<0>, grid, block, 0, 0, x, y, h, g, out, N, D, eps, p, c);
    return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? 0 : -2;
}

// out (N) = K v and / or gout (N, D) = d/dx sum_i g_i (K v)_i; either output may be NULL
int o64_kconv(int kind, const double* x, const double* y, const double* v, const double* g, double* out, double* gout, int N, int M,
              int D, double blur, const int* lab, const int* offsets, const int* intervals) {
    if (check(N, M, D) || kind < 0 || kind > 2 || (gout && !g)) return -1;
    if (N == 0) return 0;
    const Cols c{lab, offsets, intervals, M};
    const dim3 grid((N + kThreads - 1) / kThreads), block(kThreads);
    if (D == 1) hipLaunchKernelGGL(kconv_kernel<1>, grid, block, 0, 0, kind, x, y, v, g, out, gout, N, D, blur, c);
    else if (D == 2) hipLaunchKernelGGL(kconv_kernel<2>, grid, block, 0, 0, kind, x, y, v, g, out, gout, N, D, blur, c);
    else if (D == 3) hipLaunchKernelGGL(kconv_kernel<3>, grid, block, 0, 0, kind, x, y, v, g, out, gout, N, D, blur, c);
    else hipLaunchKernelGGL(kconv_kernel<0>, grid, block, 0, 0, kind, x, y, v, g, out, gout, N, D, blur, c);
    return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"
