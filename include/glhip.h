/*
 * glhip.h — C-ABI of libgeomloss_hip.so: the MI355X (gfx950) map-reduce kernels
 * behind geomloss's `SamplesLoss(backend="online"|"multiscale")`.
 *
 * The reference (jeanfeydy/geomloss 0.3.1) has no FFI of its own for this path:
 * it calls the third-party, un-vendored `pykeops` package, which JIT-compiles
 * one CUDA map-reduce per formula.  Each entry point below replaces one such
 * pykeops call site; the call site is cited next to it (paths relative to
 * /root/reference/src/geomloss/_legacy/).
 *
 * Conventions (all entry points)
 *   - plain C: raw device pointers, ints, floats; no C++ / torch types.
 *   - the caller owns every buffer; the library never allocates, frees or
 *     retains device memory.  All arrays are contiguous, row-major.
 *   - point clouds are (B, N, D) "array of structs", exactly as torch stores
 *     a contiguous (B,N,D) tensor; `in_dtype` selects their element type
 *     (GLHIP_F32 | GLHIP_BF16).  Dual vectors, weights, outputs and all
 *     accumulation are fp32.
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream) on the current device; nothing synchronises.
 *   - return value: 0 on success, a negative GLHIP_E* code otherwise; the
 *     message is available (thread-local) from glhip_last_error().
 *   - NaN/Inf propagate, they are not trapped (reference behaviour).
 *
 * Block-sparse reductions ("ranges", KeOps convention, int32 device arrays;
 * built at sinkhorn_samples.py:515 / kernel_samples.py:254-256 by
 * pykeops.torch.cluster.from_matrix):
 *   ranges_i    (n_ranges, 2)  row blocks [start, end) inside [0, N) — rows outside every
 *                              block are left untouched in the output.  Blocks are meant to be disjoint
 *                              (clusters of a sorted cloud); overlapping blocks are reduced correctly (a row
 *                              in two blocks is written by both) but one workgroup per block, without the
 *                              load-balancing row chunks
 *   slices_i    (n_ranges,)    CSR end offsets into redranges_j; block k owns
 *                              redranges_j[slices_i[k-1] : slices_i[k]] (slices_i[-1] = 0)
 *   redranges_j (nnz, 2)       column intervals [start, end) to reduce over
 *   n_ranges == 0 (pointers may be NULL) means a dense reduction over all j.
 *   Block-sparse mode requires B == 1 (as in the reference: samples_loss.py:249-257).
 *   A row block with no column interval reduces over the empty set.
 */
#ifndef GLHIP_H
#define GLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLHIP_VERSION 119 /* 0.1.19 */

/* element type of the point clouds x, y */
#define GLHIP_F32 0
#define GLHIP_BF16 1

/* kernel-convolution kinds (kernel_samples.py:62-82) */
#define GLHIP_GAUSSIAN 0  /* k = exp(-|x-y|^2 / (2 blur^2))  kernel_samples.py:62-68 */
#define GLHIP_LAPLACIAN 1 /* k = exp(-|x-y| / blur)          kernel_samples.py:71-77 */
#define GLHIP_ENERGY 2    /* k = -|x-y|  (blur ignored)      kernel_samples.py:80-82 */

/* flags (bitmask) */
#define GLHIP_FLAG_DIRECT 1   /* p=2 softmin: evaluate |x-y|^2 as sum (x_d-y_d)^2 (KeOps' SqDist)
                                 instead of the per-workgroup-centred expansion. Slower, tighter. */
#define GLHIP_FLAG_NO_MFMA 2  /* p=2 softmin / gaussian: form the exponents on the VALU instead of the matrix cores */
#define GLHIP_FLAG_NO_SPLIT 4 /* never split the columns of a row over several workgroups (ignore the workspace) */
#define GLHIP_FLAG_F32_MFMA 8 /* p=2 softmin forward: fp32 MFMA (v_mfma_f32_16x16x4_f32) instead of the bf16x3 split.  A/B only: compiled in
                                 with `make AB=1`; the shipped library runs its default kernel (same results to rounding) */
#define GLHIP_FLAG_XDL16 16   /* gaussian product: bf16x3 on 16x16x32 MFMAs (the previous tiling) instead of 32x32x16 — the tiling of the
                                 gradient kernels, see GLHIP_FLAG_GRAD_FAMILY.  p=2 softmin forward: that tiling with `make AB=1` only
                                 (A/B; the shipped library runs its default kernel) */
#define GLHIP_FLAG_GRAD_FAMILY GLHIP_FLAG_XDL16 /* kernel products: round like the product of glhip_kernel_conv_fwd_grad of the same
                                 kind — gaussian: the 16x16x32 tiling above; laplacian / energy: explicit differences with |.| = m rsq(m)
                                 (bit-identical to that product).  For the other two terms of a kernel norm whose gradient is on. */
#define GLHIP_FLAG_PREPACK 32 /* pre-pack the columns whatever the launch size (default: launches of >= 5e8 pairs); needs workspace */
#define GLHIP_FLAG_MFMA_DIST 64 /* p = 1 soft-min forward / laplacian / energy product, block-sparse launches, D <= 3: squared distances on
                                 the matrix cores, centred on each row block (glhip_dist_x32.h).  2-3x fewer VALU instructions; accurate
                                 (~2^-24 (rho + d)^2 / d on a potential, rho = row-block diameter) when row blocks are spatially compact:
                                 the caller's responsibility (voxel clusters of the multiscale backends, voxel-sorted dense clouds). */
#define GLHIP_FLAG_SMALL_ROW_BLOCKS 128 /* block-sparse soft-min forward / half-step, D <= 3: the pairs sit in row blocks of up to 64 points
                                  (sum of squared block sizes / N <= 64): launch 2-wavefront workgroups over 256-column tiles.  A hint: results
                                  do not depend on it.  The caller knows the block sizes (glhip_block_ranges_kept_pairs returns the sum). */

#define GLHIP_FLAG_F16X2 256 /* p = 2 soft-min (forward, half-step, gradient) and gaussian product / gradient, 4 <= D <= 16 (and the D <= 3
                                  soft-min forward): form the exponents from TWO f16 pieces per coordinate (3 products, v_mfma_f32_32x32x16_f16)
                                  instead of three bf16 pieces (6 products) — about half the matrix instructions and LDS bytes.  Cross terms
                                  good to ~2^-21 relative instead of 2^-24 (csrc/glhip_softmin_xd.h).  THE CALLER VOUCHES FOR THE RANGE: every
                                  exponent term (log2(e) h_j, |x - y|^2 / (2 eps ln 2)) must stay below ~2.6e5 in magnitude, i.e. roughly
                                  (cloud diameter)^2 / eps < 3e5; beyond that f16 overflows and the results are inf / nan.  Ignored by kernels
                                  without that layout (p = 1, laplacian, energy, D > 16, float64). */

#define GLHIP_FLAG_NO_SORT 512 /* p = 1 soft-min / half-step, laplacian and energy products: big dense launches (B = 1, D <= 3, N >= 65536,
                                  N M >= 5e8, a workspace of glhip_workspace_bytes) sort both clouds into the workspace themselves — voxel sort along a
                                  boustrophedon path, csrc/glhip_autosort.h — so that the squared distances can come from the matrix cores
                                  (GLHIP_FLAG_MFMA_DIST on slabs of 256 compact rows): 346 -> ~200 ms at N = M = 1e6.  Results come back in the
                                  caller's order; values differ from the unsorted launch by rounding only.  This flag keeps the clouds as they
                                  are (the generic explicit-difference kernel), e.g. for callers that hand in sorted clouds and their own ranges. */

/* Environment variables read ONCE per process by the library itself (test / tuning knobs; everything else is an argument):
 *   GLHIP_FWD_NW = 2 | 4 | 8  force the workgroup height (wavefronts) of the soft-min forward kernels instead of the size heuristic
 *   GLHIP_PREPACK_MIN = <p>   pairs per launch from which the columns are split into matrix-core records once per launch (default 5e8)
 *   GLHIP_DIST_GUARD = <x>    near-pair threshold of the matrix-core distance kernels: pairs with d^2 < x |xs_i|^2 are re-evaluated
 *                             on explicit differences (default 2^-8; 1e30 = every pair, used by the tests to check the register
 *                             <-> column map; 0 = none)
 * They are latched in function-local statics on first use: the library keeps no other global state (re-entrant apart from the
 * thread-local error string).  (Seven more A/B knobs of rounds 3-5 — GLHIP_ITER4_PRE_MIN, _ITER4_SPLITS, _TINY_MULTI_PAIRS,
 * _DIST_MULTI_MIN_COLS, _WQ_MIN_D, _XD_PRE, _H2_VIA_XD — were removed in round 6: their measured optimum is a constant now.) */

/* error codes */
#define GLHIP_OK 0
#define GLHIP_EINVAL (-1)       /* bad argument (NULL pointer, negative size, bad enum) */
#define GLHIP_EUNSUPPORTED (-2) /* valid request this build has no kernel for */
#define GLHIP_ELAUNCH (-3)      /* hipGetLastError() after the launch was not hipSuccess */

int glhip_version(void);
const char* glhip_last_error(void);

/*
 * Scratch memory.  Every reduction below accepts an optional caller-owned device buffer
 * (`workspace`, `workspace_bytes`; NULL / 0 is always valid).  With it, the columns of a row may be
 * split over several workgroups (load balance on 256 CUs) and merged by a second small kernel on the
 * same stream, and launches of >= 5e8 pairs of the p = 2 soft-min forward / gaussian product write every
 * column once as 64 bytes of bf16x3 matrix-core operands (+ 4 bytes of weight) that all row blocks then
 * copy instead of recomputing.  glhip_workspace_bytes returns a size that lets every entry point use its
 * preferred plan for the given problem (208 MB at B = 1, N = M = 1e6, D = 3); smaller buffers are used as
 * far as they go.  The buffer must stay alive until the work queued on `stream` has run; its contents
 * are scratch.
 */
size_t glhip_workspace_bytes(int B, int N, int M, int D, int n_ranges);

/*
 * Soft-C-transform  out[b,i] = -eps * log sum_j exp( h[b,j] - C(x[b,i], y[b,j]) / eps ),
 * C = |x-y|^2 / 2 (p == 2) or |x-y| (p == 1).
 *
 * Replaces: pykeops generic_logsumexp("(B - (P * cost))", ...) built by
 *   lse_genred  sinkhorn_samples.py:322-334  and  keops_lse  :432-442,
 *   as called by softmin_online :337-346, softmin_online_lazytensor :229-290 (B > 1)
 *   and softmin_multiscale :445-450 (ranges != NULL).
 * Same function as softmin_tensorized :32-71 on C = cost_routines[p](x, y) :26-29.
 *
 *   x (B,N,D)  y (B,M,D)  h (B,M) fp32  out (B,N) fp32
 *
 * Kernels by (p, D): p = 2 on the matrix cores for D <= 16 (glhip_softmin_x32.h / _xd.h; GLHIP_FLAG_F16X2 selects the two-piece f16
 * layout); p = 1 on the matrix cores for block-sparse launches of D <= 3 with GLHIP_FLAG_MFMA_DIST (glhip_dist_x32.h) and, since
 * round 5, for every DENSE launch of 4 <= D <= 16 (glhip_dist_xd.h: squared distances from the MFMA chain, pairs closer than 1/16 of
 * their offset from the cloud's centre re-evaluated exactly); everything else (D > 16, block-sparse p = 1 in D > 3, GLHIP_FLAG_NO_MFMA /
 * GLHIP_FLAG_DIRECT) on explicit differences.
 * Accuracy of the matrix-core distances (p = 1; laplacian / energy products): the squared distance of a pair carries ~2^-23 t^2 R^2
 * (t = log2(e) / eps, R = offset of the pair from the centre the launch subtracts), so the EXPONENT of a pair just above the near-pair
 * threshold d = R / 16 is off by ~2^-20 t R — it grows like diameter / eps — while out = -eps log(...) is off by
 * ~2^-20 log2(e) R <= 1.4e-6 diameter whatever eps is; a kernel VALUE of the products carries the relative error 2^-20 t R at the
 * threshold (random sign), 2^-24 t R for pairs at distance R.  Measured at eps / diameter = 0.005 and 0.002, D = 4, 5, 9
 * (tests/test_xd_kernels_gpu.py::test_distance_reductions_xd_small_blur).  GLHIP_DIST_GUARD raises the threshold.
 */
int glhip_softmin_fwd(const void* x, const void* y, const float* h, float* out,
                      int B, int N, int M, int D, float eps, int p, int in_dtype,
                      const int32_t* ranges_i, const int32_t* slices_i,
                      const int32_t* redranges_j, int n_ranges,
                      void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * One fused half-step of the symmetric Sinkhorn iteration (every kernel of D <= 3; p = 2 for D <= 16; p = 1 on dense launches for
 * D <= 16; GLHIP_EUNSUPPORTED elsewhere — compose glhip_softmin_fwd):
 *   t_i   = soft-min(eps, C(x,y), logw + pot / eps)_i          (pot == NULL: logw alone, the initialisation)
 *   out_i = damping * t_i                                       (prev == NULL)
 *   out_i = (prev_i + damping * t_i) / 2                        (prev != NULL; out must not alias prev)
 * Replaces, per soft-min call of the loop at sinkhorn_divergence.py:461-465 and :480-493, the elementwise
 * ops `log_w + pot/eps`, `damping * (...)` and `0.5 * (f + ft)` that the reference runs as separate torch kernels
 * (SURVEY §8f, N1).  Same kernels, splits and flags as glhip_softmin_fwd.
 *   logw (B,M), pot (B,M) or NULL, prev (B,N) or NULL, out (B,N): fp32.
 */
int glhip_sinkhorn_step(const void* x, const void* y, const float* logw, const float* pot, const float* prev,
                        float* out, int B, int N, int M, int D, float eps, float damping, int p, int in_dtype,
                        const int32_t* ranges_i, const int32_t* slices_i,
                        const int32_t* redranges_j, int n_ranges,
                        void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * One whole iteration of the symmetric Sinkhorn loop in ONE launch (+ one merge launch): the four
 * simultaneous updates of sinkhorn_divergence.py:480-493 (or the four initialisations of :461-465 when
 * `first` != 0), each reading the OLD potentials:
 *   f_ba' = avg(C_xy, b_log, g_ab, f_ba)    g_ab' = avg(C_yx, a_log, f_ba, g_ab)
 *   f_aa' = avg(C_xx, a_log, f_aa, f_aa)    g_bb' = avg(C_yy, b_log, g_bb, g_bb)        (debias only)
 * with avg(C, logw, pot, prev) = (prev + damping * softmin(eps, C, logw + pot/eps)) / 2  (`first` = 0),
 * damping * softmin(eps, C, logw)  (`first` = 1: initial potentials, :461-465), or
 * damping * softmin(eps, C, logw + pot/eps)  (`first` = 2: the non-averaged last update, :612-623).  C_xy = C(x_i, y_j) etc. on the clouds x (B,N,D), y (B,M,D).
 *   f_aa / g_bb and their outputs may be NULL together (debias = False: two reductions).
 * Outputs must not alias inputs.  Dense, D <= 16; p = 2 (4 <= D <= 16 since round 5) or p = 1 (round 5: the dense distance kernel
 * of glhip_dist_xd.h) (GLHIP_EUNSUPPORTED otherwise: issue four
 * glhip_sinkhorn_step calls instead); meant for small and mid-size problems, where four separate launches
 * leave the chip under-filled.  Workspace: 4 * glhip_workspace_bytes(B, max(N,M), max(N,M), D, 0).
 */
int glhip_sinkhorn_iter4(const void* x, const void* y, const float* a_log, const float* b_log,
                         const float* f_ba, const float* g_ab, const float* f_aa, const float* g_bb,
                         float* f_ba_out, float* g_ab_out, float* f_aa_out, float* g_bb_out,
                         int B, int N, int M, int D, float eps, float damping, int p, int in_dtype, int first,
                         void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * A whole run of the symmetric eps-scaling loop, queued by ONE call: the initialisation at eps[0] (sinkhorn_divergence.py:461-465)
 * followed by one averaged iteration per temperature eps[0..n_eps-1] (:468-493), each a glhip_sinkhorn_iter4 launch (+ merge) with
 * its own temperature and damping[i] — what the Python loop queues with one host-side call per iteration, minus the interpreter
 * between the launches (loops on a few thousand points are bound by the host's launch rate: N = 2000, 10 temperatures: 0.46 -> 0.36 ms).
 * The temperatures are plain host arrays, computed after whatever the caller measured (a diameter), so nothing is baked in.
 *   set0, set1: two sets of output buffers {f_ba (B,N), g_ab (B,M), f_aa (B,N), g_bb (B,M)} (the last two NULL in both sets without
 *   debiasing).  The initial potentials go to set0, iteration i reads set[i % 2] and writes set[(i + 1) % 2]: on return (queued) the
 *   final potentials are in set[n_eps % 2] and the inputs of the last iteration in the other one.  No buffer may appear twice.
 *   flags: as glhip_sinkhorn_iter4, with GLHIP_FLAG_F16X2 applied to the iterations whose eps[i] >= f16x2_min_eps only (the caller's
 *   range vouching is per temperature; pass 0 to apply it everywhere, a huge value or no flag for never).
 * Dense, D <= 16, p = 1 or 2, n_eps >= 1; workspace as glhip_sinkhorn_iter4.  Used for single-scale losses and for the coarse level of
 * the two-scale ones (everything up to the jump).
 */
int glhip_sinkhorn_anneal(const void* x, const void* y, const float* a_log, const float* b_log, float* const* set0, float* const* set1,
                          int B, int N, int M, int D, const float* eps, const float* damping, int n_eps, int p, int in_dtype,
                          void* workspace, size_t workspace_bytes, int flags, float f16x2_min_eps, void* stream);

/*
 * The coarse-to-fine jump of the two-scale loop in ONE launch (+ one merge launch): the four extrapolations of
 * sinkhorn_divergence.py:590-599, each `extrapolate_samples` (sinkhorn_samples.py:533-544) — one soft-min of the FINE points against
 * a COARSE measure:
 *   f_ba'(x_i) = damping * softmin(eps, C(x_i, yc_j), b_log_c + g_ab / eps)    g_ab'(y_i) = damping * softmin(eps, C(y_i, xc_j), a_log_c + f_ba / eps)
 *   f_aa'(x_i) = damping * softmin(eps, C(x_i, xc_j), a_log_c + f_aa / eps)    g_bb'(y_i) = damping * softmin(eps, C(y_i, yc_j), b_log_c + g_bb / eps)
 * x (B,N,D), y (B,M,D): the fine clouds; xc (B,Nc,D), yc (B,Mc,D): the coarse ones (centroids), with their log-weights a_log_c (B,Nc),
 * b_log_c (B,Mc) and potentials f_ba, f_aa (B,Nc), g_ab, g_bb (B,Mc); outputs (B,N) / (B,M), all fp32.  f_aa / g_bb and their outputs
 * may be NULL together.  Same kernels, flags, limits (dense, D <= 16, p = 1 or 2) and error behaviour as glhip_sinkhorn_iter4, of which
 * this is the rows != columns form; replaces four glhip_softmin_fwd launches and their 12 elementwise torch kernels per jump.
 * Workspace: 4 * glhip_workspace_bytes(B, max(N, M), max(Nc, Mc), D, 0) (rows: the fine clouds, columns: the coarse ones); as with every
 * entry point a smaller (or NULL) workspace is accepted: fewer column splits, no pre-packed columns, same results up to rounding.
 */
int glhip_sinkhorn_extrapolate4(const void* x, const void* y, const void* xc, const void* yc, const float* a_log_c, const float* b_log_c,
                                const float* f_ba, const float* g_ab, const float* f_aa, const float* g_bb,
                                float* f_ba_out, float* g_ab_out, float* f_aa_out, float* g_bb_out,
                                int B, int N, int M, int Nc, int Mc, int D, float eps, float damping, int p, int in_dtype,
                                void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * The elementwise front and back end of a Sinkhorn loss, for problems of a few thousand points whose time is the host's launch rate
 * (round 6; a 2000-point loss is 26 launches of which the soft-mins are 9):
 *   glhip_log_weights: out_k[i] = log(w_k[i]) with log(0) -> -100000 for up to 4 vectors in ONE launch — `log_weights`
 *     (sinkhorn_divergence.py:61-65; NaN weights stay NaN).  w, out, n: HOST arrays of `count` device pointers / lengths.
 *   glhip_sinkhorn_cost: out[b] = <a_b, f_ba_b - f_aa_b> + <b_b, g_ab_b - g_bb_b> — `sinkhorn_cost`, balanced case
 *     (sinkhorn_divergence.py:171-199) — accumulated in float64 in a fixed order; f_aa = g_bb = NULL: without debiasing.
 *     f_*, (B,N) and g_*, (B,M) fp32; a (B,N) if a_batched else (N) shared by the batch, b likewise; out (B) fp32.  One workgroup per
 *     batch item: meant for N, M up to a few 1e4 (beyond, a tree of torch reductions is faster).
 *   glhip_bounding_box: lo_hi (2 D) fp32 out = the D minima then the D maxima of the coordinates of x (nx, D) and y (ny, D) together
 *     (either may be empty: +inf / -inf there) — the reductions of `max_diameter` (sinkhorn_divergence.py:96-112: cat, aminmax) in one
 *     launch of one workgroup; exact (minima and maxima); D <= 16; meant for clouds of up to ~1e4 points (one workgroup sweeps them).
 */
int glhip_bounding_box(const void* x, long nx, const void* y, long ny, int D, int in_dtype, float* lo_hi, void* stream);
int glhip_log_weights(const float* const* w, float* const* out, const long* n, int count, void* stream);
int glhip_sinkhorn_cost(const float* a, const float* f_ba, const float* f_aa, const float* b, const float* g_ab, const float* g_bb,
                        float* out, int B, int N, int M, int a_batched, int b_batched, void* stream);

/*
 * Log-sum-exp along the lines of a regular grid — the separable soft-min of the reference's image / volume path
 * (SURVEY §8f, N4):
 *   out[r, i] = log sum_j exp( h[r, j] - c(i, j) ),   c(i, j) = (x_i - x_j)^2 / (2 eps)  (p = 2)  or  |x_i - x_j| / eps  (p = 1),
 *   x_i = i / N, for R independent lines of N <= 4096 fp32 samples stored contiguously.
 * Replaces: the KeOps `LazyTensor.logsumexp(dim=2)` inside `softmin_grid` (_legacy/utils.py:254-270), which the
 * reference applies once per image axis (:272-283; the caller permutes the axis of interest to the last position).
 * glhip_lse_lines_bwd is its vector-Jacobian product:  grad_h[r, j] = sum_i grad_out[r, i] exp(h[r, j] - c(i, j) - lse[r, i]).
 */
int glhip_lse_lines_fwd(const float* h, float* out, long R, int N, float eps, int p, void* stream);
int glhip_lse_lines_bwd(const float* h, const float* lse, const float* grad_out, float* grad_h, long R, int N,
                        float eps, int p, void* stream);

/*
 * Gradient of glhip_softmin_fwd with respect to x (the only differentiable argument on the
 * reference's path: y and h are detached at sinkhorn_samples.py:392-393,628-651 and
 * sinkhorn_divergence.py:616-623):
 *   grad_x[b,i,:] = grad_out[b,i] * sum_j P_ij * dC/dx(x_i, y_j),
 *   P_ij = exp( h_j - C_ij/eps + out_i/eps )   (rows of P sum to 1; we renormalise by the
 *   recomputed row sum, as autograd's logsumexp backward does).
 * Replaces: the symbolic KeOps `Grad` of the generic_logsumexp above.
 * (p = 1, dense, 4 <= D <= 16: glhip_dist_xd.h since round 5 — distances from the MFMA chain, near pairs from the points themselves.)
 *   out = the saved forward result (B,N);  grad_out (B,N) fp32;  grad_x (B,N,D) fp32.
 */
int glhip_softmin_bwd_x(const void* x, const void* y, const float* h,
                        const float* out, const float* grad_out, float* grad_x,
                        int B, int N, int M, int D, float eps, int p, int in_dtype,
                        const int32_t* ranges_i, const int32_t* slices_i,
                        const int32_t* redranges_j, int n_ranges,
                        void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * Kernel-matrix × vector product  out[b,i] = sum_j k(x[b,i], y[b,j]) * v[b,j].
 *
 * Replaces: `K @ v` on a KeOps LazyTensor K (kernel_samples.py:128,130,132,136-137)
 *   with K from gaussian_kernel / laplacian_kernel / energy_kernel (:62-82),
 *   optionally block-sparse (K.ranges = ranges, :66-67,75-76).
 *   The transposed product K^T @ a (:135-137) is the same call with x and y swapped.
 *
 *   v (B,M) fp32, out (B,N) fp32.
 *   gaussian: matrix cores for D <= 16; laplacian / energy: matrix-core distances for block-sparse D <= 3 launches with
 *   GLHIP_FLAG_MFMA_DIST and for dense launches of 4 <= D <= 16 (glhip_dist_xd.h, round 5); explicit differences elsewhere.
 */
int glhip_kernel_conv_fwd(int kind, const void* x, const void* y, const float* v, float* out,
                          int B, int N, int M, int D, float blur, int in_dtype,
                          const int32_t* ranges_i, const int32_t* slices_i,
                          const int32_t* redranges_j, int n_ranges,
                          void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * Gradient of sum_i g_i * out_i (out from glhip_kernel_conv_fwd) with respect to x:
 *   grad_x[b,i,:] = g[b,i] * sum_j v[b,j] * dk/dx(x_i, y_j),
 *   with the convention d|z|/dz = 0 at z = 0 (what KeOps' Norm2 gradient returns and what the
 *   clamp at utils.py:61 yields in the dense code).
 * The gradient with respect to y is the same call with (x,g) and (y,v) swapped; the gradient
 * with respect to v is glhip_kernel_conv_fwd with x and y swapped and v := g.
 * Replaces: KeOps autograd of the reductions at kernel_samples.py:128-137.
 */
int glhip_kernel_conv_bwd_x(int kind, const void* x, const void* y, const float* v,
                            const float* g, float* grad_x,
                            int B, int N, int M, int D, float blur, int in_dtype,
                            const int32_t* ranges_i, const int32_t* slices_i,
                            const int32_t* redranges_j, int n_ranges,
                            void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * Row-wise soft-min of an explicit (B,N,M) fp32 cost matrix:
 *   out[b,i] = -eps * log sum_j exp( h[b,j] - C[b,i,j] / eps ).
 * Replaces: softmin_tensorized sinkhorn_samples.py:32-71 for CUDA tensors
 *   (the `backend="tensorized"` path, including user-supplied `cost` callables).
 */
int glhip_softmin_dense_fwd(const float* C, const float* h, float* out,
                            int B, int N, int M, float eps, void* stream);

/*
 * Soft-min AND its gradient with respect to the row points in ONE pass (p = 2, D <= 16), for callers that know the answer
 * approximately — the last, differentiable update of the Sinkhorn loop (sinkhorn_divergence.py:612-623), whose previous iterate
 * bounds it: the soft-min is 1-Lipschitz in its dual vector, so |out - guess| <= margin := sup_j |h_j - h_j(previous)| * eps.
 *   guess (B,N): soft-min values for the previous dual vector;  margin >= 0 (natural units, like out).
 *   out[b,i]          = -eps log sum_j exp(h_j - C_ij / eps)                      (exact, whatever the guess)
 *   grad_unit[b,i,:]  = d out[b,i] / d x[b,i,:] = sum_j P_ij (x_i - y_j)
 * The weights are formed relative to guess + margin (an upper bound of out), so none exceeds 1 and their sum is >= exp(-2 margin/eps):
 * keep margin / eps below ~25 (the caller checks; beyond that run glhip_softmin_fwd + glhip_softmin_bwd_x).  Rounding: out is
 * guess plus a correction of up to 2 margins, so it carries a few ulps of the margin (<= 4e-7 margin, measured by
 * tools/fuzz_kernels.py) on top of the error of glhip_softmin_fwd.  One reduction of the cost of glhip_softmin_bwd_x instead of a
 * forward plus a backward reduction.
 */
int glhip_softmin_fwd_grad(const void* x, const void* y, const float* h, const float* guess, float margin, float* out,
                           float* grad_unit, int B, int N, int M, int D, float eps, int p, int in_dtype,
                           const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges,
                           void* workspace, size_t workspace_bytes, int flags, void* stream);
/*
 * Kernel product AND its gradient with respect to the row points in ONE pass (gaussian kernel, D <= 3):
 *   out[b,i]         = sum_j k(x_i, y_j) v_j                      (what glhip_kernel_conv_fwd returns)
 *   grad_unit[b,i,:] = d out[b,i] / d x[b,i,:] = -(1/blur^2) sum_j v_j k(x_i, y_j) (x_i - y_j)
 * i.e. glhip_kernel_conv_bwd_x for grad_out = 1 — the mass accumulator of that reduction IS the product.  The autograd
 * forward of the kernel norms calls this when x requires gradients; the backward pass is then grad_out[b,i] * grad_unit[b,i,:],
 * an elementwise product (kernel_samples.py:92-146 with gradients: 3 + 2 reductions become 3).
 * gaussian: matrix-core kernel (GLHIP_FLAG_NO_MFMA: explicit differences).  laplacian / energy: explicit differences with
 * |.| = m rsq(m); the product is then bit-identical to glhip_kernel_conv_fwd under GLHIP_FLAG_GRAD_FAMILY, which is what the
 * three terms of one kernel norm are sent to when gradients are on (their rounding must be common to cancel).  Coincident points (inside the 1e-4
 * clamp of utils.py:61) add k(0) v_j to the product and nothing to the gradient.  D > 3: GLHIP_EUNSUPPORTED (call the two
 * entry points above).
 */
int glhip_kernel_conv_fwd_grad(int kind, const void* x, const void* y, const float* v, float* out, float* grad_unit,
                               int B, int N, int M, int D, float blur, int in_dtype,
                               const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges,
                               void* workspace, size_t workspace_bytes, int flags, void* stream);
/*
 * Hard C-transforms — the eps -> 0 limits of the two soft-mins above (min / max reductions, no exponential).
 *   glhip_cmin_fwd:       out[b,i] = min_j [ C(x[b,i], y[b,j]) - g[b,j] ],  C = |x-y|^2/2 (p = 2) or |x-y| (p = 1), D <= 3;
 *                         +inf over an empty column set.  Replaces the `eps == 0` branch of softmin_sample
 *                         (ot/_implementations/sample.py:156-166: `(C_xy - g_y_j).min(axis=1)` on a LazyTensor).
 *   glhip_max_lines_fwd:  out[r,i] = max_j [ g[r,j] - c(i,j) ],  c = (step (i-j))^2 (p = 2) or step |i-j| (p = 1), for R lines
 *                         of N <= 4096 samples.  Replaces the LazyTensor `.max(dim=2)` of `C_transform`
 *                         (_legacy/utils.py:116-182), applied once per image axis.
 */
int glhip_cmin_fwd(const void* x, const void* y, const float* g, float* out, int B, int N, int M, int D, int p, int in_dtype,
                   const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges,
                   void* workspace, size_t workspace_bytes, int flags, void* stream);
int glhip_max_lines_fwd(const float* g, float* out, long R, int N, float step, int p, void* stream);
/*
 * The cluster pyramid of the two-scale ("multiscale") backends, on the device.
 *
 * glhip_grid_cluster — voxel clustering of a weighted cloud.  Replaces the chain
 *   pykeops.torch.cluster.grid_cluster -> cluster_ranges_centroids -> sort_clusters
 * of `clusterize`, sinkhorn_samples.py:453-490, and kernel_samples.py:214-236: points are binned in cubic voxels of edge
 * `voxel` (bin of a coordinate c: floor((c / pre_div) / voxel); kernel_multiscale clusters x / blur, sinkhorn passes 1),
 * clusters are numbered in lexicographic voxel order (first axis most significant), and the cloud is sorted by cluster with
 * a STABLE sort, so that cluster k is the contiguous row range [ranges[2k], ranges[2k+1]) of the sorted cloud.
 *   x (N,D) fp32|bf16, D <= 3;  weights (N) fp32 or NULL (all ones)
 *   perm (N) int32 out: sorted row p is input row perm[p];   x_sorted (N,D) / w_sorted (N) out, may be NULL
 *   ranges (N,2) int32, centroids (N,D) fp32 (weighted means of x / pre_div), weights_c (N) fp32: the first C rows are
 *   written, C <= N = the number of non-empty voxels;  n_clusters (8) int32 out: {C, overflow, qmin[3], qmax[3]} — overflow != 0
 *   means a voxel coordinate exceeded 2^21 bins along an axis (result invalid; use a larger voxel); qmin / qmax are the smallest
 *   and largest voxel index of the cloud along each axis (0 beyond D; since version 114): the cloud lies in the box
 *   [qmin voxel pre_div, (qmax + 1) voxel pre_div) — its bounding box to one voxel, which is how a caller that was GIVEN a
 *   diameter (it only parametrises the schedule, _legacy/sinkhorn_divergence.py:154-163) learns the true extent of the data
 *   before vouching for GLHIP_FLAG_F16X2, in the round trip that reads C anyway.
 * Sums are accumulated in float64 in a fixed order: the same inputs give the same centroids bit for bit.  Nothing comes
 * back to the host: read n_clusters when the sizes are needed.  workspace: glhip_cluster_workspace_bytes(N, D).
 *
 * glhip_block_ranges — a keep rule on pairs of clusters -> block-sparse reduction ranges for both orientations.
 * Replaces `keep = ...` + pykeops.torch.cluster.from_matrix at sinkhorn_samples.py:512-530 (GLHIP_KEEP_DUAL_SLACK:
 * keep (i,j) iff f_i + g_j > C(rows_i, cols_j) - thr, C the cost of exponent p on the centroids, thr = truncate * eps) and
 * kernel_samples.py:244-256 (GLHIP_KEEP_WITHIN: keep iff |rows_i - cols_j|^2 <= thr, thr = (truncate + cell diameter)^2).
 *   rows (Cr,D), cols (Cc,D) fp32 centroids;  f (Cr), g (Cc) dual values (dual slack only);
 *   ranges_rows (Cr,2), ranges_cols (Cc,2): row ranges of the clusters in their sorted clouds;
 *   out: slices_rows (Cr) + red_cols (capacity,2) = the (slices_i, redranges_j) of a reduction over the columns of every
 *        row cluster, slices_cols (Cc) + red_rows (capacity,2) = the same for the transposed reduction.  Kept column
 *        clusters that are adjacent in memory are merged into one interval (same pair set, fewer and longer tiles).
 *   capacity: intervals each `red_*` array can hold (64-bit); Cr * ((Cc + 1) / 2) (resp. Cc * ((Cr + 1) / 2)) always suffices,
 *             glhip_block_ranges_count gives the exact number.
 *   status (1) int32 out: != 0 if capacity was exceeded (intervals beyond it are dropped, never written out of bounds).
 *   slices_cols == red_rows == NULL (since version 117): the row-major pattern only — for symmetric patterns (rows = cols, f = g: the
 *   debiasing terms C_xx, C_yy of sinkhorn_divergence.py:284-289), whose transposed pattern is the same arrays.
 */
#define GLHIP_KEEP_DUAL_SLACK 0
#define GLHIP_KEEP_WITHIN 1
size_t glhip_cluster_workspace_bytes(int N, int D);
int glhip_grid_cluster(const void* x, const float* weights, int N, int D, int in_dtype, float pre_div, float voxel,
                       int32_t* perm, void* x_sorted, float* w_sorted, int32_t* ranges, float* centroids,
                       float* weights_c, int32_t* n_clusters, void* workspace, size_t workspace_bytes, void* stream);
int glhip_block_ranges(int kind, const float* rows, const float* cols, const float* f, const float* g, int Cr, int Cc,
                       int D, int p, float thr, const int32_t* ranges_rows, const int32_t* ranges_cols,
                       int32_t* slices_rows, int32_t* red_cols, int32_t* slices_cols, int32_t* red_rows, long long capacity,
                       int32_t* status, void* stream);
/* The counting half of glhip_block_ranges alone: writes the CSR offsets slices_rows (Cr) / slices_cols (Cc) and
 * totals (2) = { number of intervals of the row-major pattern, of the column-major pattern } (= the last offsets).  A caller that
 * reads `totals` back can size red_cols / red_rows exactly instead of for the worst case (which is quadratic in the number of
 * clusters: 3.2 GB at 2e4 clusters a side) and then call glhip_block_ranges with capacity = max(totals). */
int glhip_block_ranges_count(int kind, const float* rows, const float* cols, const float* f, const float* g, int Cr, int Cc,
                             int D, int p, float thr, const int32_t* ranges_rows, const int32_t* ranges_cols,
                             int32_t* slices_rows, int32_t* slices_cols, int32_t* totals, void* stream);

/* Pairs of POINTS the keep rule of glhip_block_ranges retains (same arguments), without building the intervals:
 * kept (3) int64 out = { sum over the kept cluster pairs (i, j) of |rows_i| x |cols_j|,  sum_i |rows_i|^2,  sum_j |cols_j|^2 }.
 * The reference prints the first figure at the level of clusters when verbose (sinkhorn_samples.py:516-522); the host side costs a
 * block-sparse fine level against a dense one with it BEFORE building the pattern (geomloss_amd/sinkhorn_samples.py:
 * kernel_truncation; `truncate=None` at :504-505 is the dense one).  The other two give the size of the block a typical PAIR lives in
 * (sum of squares / number of points), which is what GLHIP_FLAG_SMALL_ROW_BLOCKS is set from. */
int glhip_block_ranges_kept_pairs(int kind, const float* rows, const float* cols, const float* f, const float* g, int Cr, int Cc,
                                  int D, int p, float thr, const int32_t* ranges_rows, const int32_t* ranges_cols, long long* kept,
                                  void* stream);

/* ---- the four reductions in DOUBLE precision (round 4) ---------------------------------------------------------------------
 * The reference's matrix-free backends keep the dtype of their inputs: float64 clouds are reduced in float64 by KeOps
 * (softmin_online_lazytensor, sinkhorn_samples.py:229-290, `.logsumexp` on LazyTensors of the input dtype; lse_genred :322-334
 * with dtype = float64; kernel_online, kernel_samples.py:128-137).  These entry points are that path: every array — clouds, dual
 * vector / weights, gradients, outputs — is `double`; same argument meaning, ranges convention and semantics (clamp of
 * utils.py:61, zero direction at clamped pairs) as glhip_softmin_fwd / glhip_softmin_bwd_x / glhip_kernel_conv_fwd /
 * glhip_kernel_conv_bwd_x above; any D up to 4095 (D > 16 since round 5: run-time coordinate loops, slower); no workspace, no flags.
 * 1 to 64 threads per row, explicit differences, no matrix cores: ~5e11 pairs/s, for callers who need the digits.  Of the fused entry points
 * only the half-step has a float64 form (below); iteration and value + gradient: compose these. */
int glhip_softmin_fwd_f64(const double* x, const double* y, const double* h, double* out, int B, int N, int M, int D, double eps, int p,
                          const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges, void* stream);
/* glhip_sinkhorn_step in double precision (round 6): out_i = damping * t_i (prev == NULL) or (prev_i + damping * t_i) / 2, with
 * t = soft-min(eps, C(x,y), logw + pot / eps) (pot == NULL: logw alone); out must not alias prev.  One launch instead of the soft-min
 * and five elementwise float64 torch kernels per half-step of sinkhorn_divergence.py:461-465,480-493. */
int glhip_sinkhorn_step_f64(const double* x, const double* y, const double* logw, const double* pot, const double* prev, double* out,
                            int B, int N, int M, int D, double eps, double damping, int p, const int32_t* ranges_i,
                            const int32_t* slices_i, const int32_t* redranges_j, int n_ranges, void* stream);
int glhip_softmin_bwd_x_f64(const double* x, const double* y, const double* h, const double* out, const double* grad_out, double* grad_x,
                            int B, int N, int M, int D, double eps, int p, const int32_t* ranges_i, const int32_t* slices_i,
                            const int32_t* redranges_j, int n_ranges, void* stream);
int glhip_kernel_conv_fwd_f64(int kind, const double* x, const double* y, const double* v, double* out, int B, int N, int M, int D,
                              double blur, const int32_t* ranges_i, const int32_t* slices_i, const int32_t* redranges_j, int n_ranges,
                              void* stream);
int glhip_kernel_conv_bwd_x_f64(int kind, const double* x, const double* y, const double* v, const double* grad_out, double* grad_x, int B,
                                int N, int M, int D, double blur, const int32_t* ranges_i, const int32_t* slices_i,
                                const int32_t* redranges_j, int n_ranges, void* stream);
#ifdef __cplusplus
}
#endif
#endif /* GLHIP_H */
