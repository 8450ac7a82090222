"""CPU oracle (NumPy, float64 by default) for the hot path of geomloss's ``SamplesLoss``.

TEST INFRASTRUCTURE ONLY.  Nothing under ``geomloss_amd/`` imports this module; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.

Every function restates one piece of the reference (paths relative to
/root/reference/src/geomloss/_legacy/) and cites it.  Pinning: the reference's own test-suite
never exercises ``SamplesLoss``; this oracle is pinned instead against outputs of the reference's
tensorized backend generated in the build container (``tests/golden/*.npz``, produced by
``tests/golden/make_golden.py``) — see ``tests/test_oracle_golden.py``.  The two-scale
("multiscale") algorithm cannot be run from the reference here (it needs pykeops, which is not
installed and not vendored): for that backend the oracle restates the algorithm of
``sinkhorn_samples.py:453-681`` with dense masked matrices, and its clustering follows the
documented semantics of ``pykeops.torch.cluster`` — parity of the *clustering helper* with
pykeops itself is unpinned.  The grid / image path (``utils.py:64-279``, ``sinkhorn_images.py``,
``wasserstein_barycenter_images.py``) is pinned against the reference's own files, run with a
dense stand-in for the one pykeops primitive they call (``tests/golden/make_golden_images.py``,
``tests/test_images_cpu.py``: agreement <= 1e-15).
"""

import numpy as np

# --------------------------------------------------------------------------------------------------
#  costs and the soft-min
# --------------------------------------------------------------------------------------------------


def squared_distances(x, y):
    """|x_i - y_j|^2 for (..., N, D) x (..., M, D), by explicit differences (utils.py:26-53 computes the
    same quantity through |x|^2 - 2x.y + |y|^2; in float64 the two agree to ~1e-15)."""
    diff = x[..., :, None, :] - y[..., None, :, :]
    return (diff * diff).sum(-1)


def distances(x, y):
    """sqrt(max(|x_i - y_j|^2, 1e-8))  — utils.py:56-61 (dense mode clamps before the root)."""
    return np.sqrt(np.maximum(squared_distances(x, y), 1e-8))


def cost_matrix(x, y, p=2):
    """C(x_i, y_j) = |x_i-y_j|^2 / 2 (p=2) or |x_i-y_j| (p=1)  — cost_routines, sinkhorn_samples.py:26-29."""
    if p == 2:
        return squared_distances(x, y) / 2
    if p == 1:
        return distances(x, y)
    raise NotImplementedError("p must be 1 or 2")


def logsumexp(v, axis=-1):
    m = np.max(v, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide="ignore"):
        return np.log(np.exp(v - m).sum(axis=axis)) + np.squeeze(m, axis=axis)


def softmin_dense(eps, C, h):
    """f_i = -eps * log sum_j exp(h_j - C_ij/eps)  — softmin_tensorized, sinkhorn_samples.py:70-71.
    C: (..., N, M), h: (..., M) -> (..., N)."""
    return -eps * logsumexp(h[..., None, :] - C / eps, axis=-1)


def softmin_points(eps, x, y, h, p=2, dtype=np.float64, row_block=2048):
    """Same quantity from the points, row-blocked so that large N x M fit in memory
    (what softmin_online computes: sinkhorn_samples.py:337-346)."""
    x, y, h = np.asarray(x, dtype), np.asarray(y, dtype), np.asarray(h, dtype)
    out = np.empty(x.shape[:-1], dtype)
    N = x.shape[-2]
    for i0 in range(0, N, row_block):
        C = cost_matrix(x[..., i0:i0 + row_block, :], y, p)
        out[..., i0:i0 + row_block] = softmin_dense(eps, C, h)
    return out


def softmin_points_grad_x(eps, x, y, h, g, p=2, dtype=np.float64, row_block=2048):
    """d/dx of sum_i g_i f_i:  g_i sum_j P_ij dC/dx(x_i,y_j), P = softmax_j(h_j - C_ij/eps).
    (What autograd produces through logsumexp in the tensorized code; SURVEY Appendix A.)"""
    x, y, h, g = (np.asarray(t, dtype) for t in (x, y, h, g))
    gx = np.empty_like(x)
    N = x.shape[-2]
    for i0 in range(0, N, row_block):
        xb = x[..., i0:i0 + row_block, :]
        C = cost_matrix(xb, y, p)
        v = h[..., None, :] - C / eps
        v = v - v.max(-1, keepdims=True)
        P = np.exp(v)
        P /= P.sum(-1, keepdims=True)
        diff = xb[..., :, None, :] - y[..., None, :, :]
        if p == 1:  # gradient of sqrt(clamp_min(d2, 1e-8)): zero where the clamp is active
            n2 = (diff * diff).sum(-1, keepdims=True)
            diff = np.where(n2 > 1e-8, diff / np.sqrt(np.where(n2 > 1e-8, n2, 1.0)), 0.0)
        gx[..., i0:i0 + row_block, :] = g[..., i0:i0 + row_block, None] * (P[..., None] * diff).sum(-2)
    return gx


# --------------------------------------------------------------------------------------------------
#  epsilon schedule and loss formulas  (sinkhorn_divergence.py)
# --------------------------------------------------------------------------------------------------


def dampening(eps, rho):
    """sinkhorn_divergence.py:56-58"""
    return 1.0 if rho is None else 1.0 / (1.0 + eps / rho)


def log_weights(a):
    """sinkhorn_divergence.py:61-65"""
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.log(a)
    out[a <= 0] = -100000.0
    return out


def max_diameter(x, y):
    """sinkhorn_divergence.py:96-112 (bounding box of the flattened batch, :156-158)"""
    D = x.shape[-1]
    xf, yf = x.reshape(-1, D), y.reshape(-1, D)
    mins = np.minimum(xf.min(0), yf.min(0))
    maxs = np.maximum(xf.max(0), yf.max(0))
    return float(np.linalg.norm(maxs - mins))


def epsilon_schedule(p, diameter, blur, scaling):
    """sinkhorn_divergence.py:141-151"""
    return ([diameter**p]
            + [np.exp(e) for e in np.arange(p * np.log(diameter), p * np.log(blur), p * np.log(scaling))]
            + [blur**p])


def scaling_parameters(x, y, p, blur, reach, diameter, scaling):
    """sinkhorn_divergence.py:154-163"""
    if diameter is None:
        # the reference measures the box on the tensors it is given (float32 there): do the same
        diameter = max_diameter(x.astype(np.float32), y.astype(np.float32))
    eps = blur**p
    rho = None if reach is None else reach**p
    return diameter, eps, epsilon_schedule(p, diameter, blur, scaling), rho


def scal(a, f):
    """utils.py:13-18, batched over leading axes"""
    return (a * f).sum(-1)


def sinkhorn_cost(eps, rho, a, b, f_aa, g_bb, g_ab, f_ba, debias=True, potentials=False):
    """sinkhorn_divergence.py:171-250"""
    if potentials:
        return (f_ba - f_aa, g_ab - g_bb) if debias else (f_ba, g_ab)
    if rho is None:
        if debias:
            return scal(a, f_ba - f_aa) + scal(b, g_ab - g_bb)
        return scal(a, f_ba) + scal(b, g_ab)
    w = rho + eps / 2
    if debias:
        return (scal(a, w * (np.exp(-f_aa / rho) - np.exp(-f_ba / rho)))
                + scal(b, w * (np.exp(-g_bb / rho) - np.exp(-g_ab / rho))))
    return scal(a, w * (1 - np.exp(-f_ba / rho))) + scal(b, w * (1 - np.exp(-g_ab / rho)))


# --------------------------------------------------------------------------------------------------
#  the Sinkhorn loop  (sinkhorn_divergence.py:258-628)
# --------------------------------------------------------------------------------------------------


def sinkhorn_loop(softmin, a_logs, b_logs, C_xxs, C_yys, C_xys, C_yxs, eps_list, rho, jumps=(),
                  kernel_truncation=None, truncate=5, cost=None, extrapolate=None, debias=True):
    """Returns (f_aa, g_bb, g_ab, f_ba) and the inputs of the last (differentiable) soft-mins.

    Level lists have one entry (single scale) or two (coarse, fine).  Statement order follows
    sinkhorn_divergence.py:434-628: init at eps_list[0] (:461-465), symmetric averaged updates
    (:468-493), optional jump with truncation + extrapolation (:519-606), last non-averaged update
    (:612-623)."""
    k = 0
    a_log, b_log = a_logs[k], b_logs[k]
    C_xy, C_yx = C_xys[k], C_yxs[k]
    C_xx, C_yy = (C_xxs[k], C_yys[k]) if debias else (None, None)

    eps = eps_list[0]
    lam = dampening(eps, rho)
    g_ab = lam * softmin(eps, C_yx, a_log)
    f_ba = lam * softmin(eps, C_xy, b_log)
    if debias:
        f_aa = lam * softmin(eps, C_xx, a_log)
        g_bb = lam * softmin(eps, C_yy, b_log)
    else:
        f_aa = g_bb = None

    last_extrapolation = True
    for i, eps in enumerate(eps_list):
        lam = dampening(eps, rho)
        ft_ba = lam * softmin(eps, C_xy, b_log + g_ab / eps)
        gt_ab = lam * softmin(eps, C_yx, a_log + f_ba / eps)
        if debias:
            ft_aa = lam * softmin(eps, C_xx, a_log + f_aa / eps)
            gt_bb = lam * softmin(eps, C_yy, b_log + g_bb / eps)
        f_ba, g_ab = 0.5 * (f_ba + ft_ba), 0.5 * (g_ab + gt_ab)
        if debias:
            f_aa, g_bb = 0.5 * (f_aa + ft_aa), 0.5 * (g_bb + gt_bb)

        if i in jumps:
            if i == len(eps_list) - 1:
                C_xy_f, C_yx_f = C_xys[k + 1], C_yxs[k + 1]
                if debias:
                    C_xx_f, C_yy_f = C_xxs[k + 1], C_yys[k + 1]
                last_extrapolation = False
            else:
                C_xy_f, C_yx_f = kernel_truncation(C_xy, C_yx, C_xys[k + 1], C_yxs[k + 1], f_ba, g_ab, eps,
                                                   truncate=truncate, cost=cost)
                if debias:
                    C_xx_f, _ = kernel_truncation(C_xx, C_xx, C_xxs[k + 1], C_xxs[k + 1], f_aa, f_aa, eps,
                                                  truncate=truncate, cost=cost)
                    C_yy_f, _ = kernel_truncation(C_yy, C_yy, C_yys[k + 1], C_yys[k + 1], g_bb, g_bb, eps,
                                                  truncate=truncate, cost=cost)
            f_ba, g_ab = (extrapolate(f_ba, g_ab, eps, lam, C_xy, b_log, C_xy_f),
                          extrapolate(g_ab, f_ba, eps, lam, C_yx, a_log, C_yx_f))
            if debias:
                f_aa = extrapolate(f_aa, f_aa, eps, lam, C_xx, a_log, C_xx_f)
                g_bb = extrapolate(g_bb, g_bb, eps, lam, C_yy, b_log, C_yy_f)
            k += 1
            a_log, b_log = a_logs[k], b_logs[k]
            C_xy, C_yx = C_xy_f, C_yx_f
            if debias:
                C_xx, C_yy = C_xx_f, C_yy_f

    last = dict(eps=eps, lam=lam, C_xy=C_xy, C_yx=C_yx, C_xx=C_xx, C_yy=C_yy, a_log=a_log, b_log=b_log)
    if last_extrapolation:
        last["h_ba"], last["h_ab"] = b_log + g_ab / eps, a_log + f_ba / eps
        if debias:
            last["h_aa"], last["h_bb"] = a_log + f_aa / eps, b_log + g_bb / eps
        f_ba, g_ab = lam * softmin(eps, C_xy, last["h_ba"]), lam * softmin(eps, C_yx, last["h_ab"])
        if debias:
            f_aa, g_bb = lam * softmin(eps, C_xx, last["h_aa"]), lam * softmin(eps, C_yy, last["h_bb"])
    return (f_aa, g_bb, g_ab, f_ba), last


# --------------------------------------------------------------------------------------------------
#  tensorized Sinkhorn  (sinkhorn_samples.py:74-221)  = the parity target of BASELINE.json
# --------------------------------------------------------------------------------------------------


def _uniform(n, lead, dtype):
    return np.full(lead + (n,), 1.0 / n, dtype)


def sinkhorn_tensorized(a, x, b, y, p=2, blur=0.05, reach=None, diameter=None, scaling=0.5, debias=True,
                        potentials=False, return_internals=False):
    """a (...,N), x (...,N,D), b (...,M), y (...,M,D) -> loss (...) or potentials."""
    C_xy, C_yx = cost_matrix(x, y, p), cost_matrix(y, x, p)
    C_xx, C_yy = (cost_matrix(x, x, p), cost_matrix(y, y, p)) if debias else (None, None)
    diameter, eps, eps_list, rho = scaling_parameters(x, y, p, blur, reach, diameter, scaling)
    pots, last = sinkhorn_loop(softmin_dense, [log_weights(a)], [log_weights(b)], [C_xx], [C_yy], [C_xy], [C_yx],
                               eps_list, rho, debias=debias)
    f_aa, g_bb, g_ab, f_ba = pots
    out = sinkhorn_cost(eps, rho, a, b, f_aa, g_bb, g_ab, f_ba, debias=debias, potentials=potentials)
    if return_internals:
        return out, pots, last, (eps, rho, eps_list)
    return out


def sinkhorn_loss(x, y, a=None, b=None, dtype=np.float64, **kw):
    """SamplesLoss("sinkhorn", backend="tensorized")(a, x, b, y) for float64 NumPy inputs."""
    x, y = np.asarray(x, dtype), np.asarray(y, dtype)
    a = _uniform(x.shape[-2], x.shape[:-2], dtype) if a is None else np.asarray(a, dtype)
    b = _uniform(y.shape[-2], y.shape[:-2], dtype) if b is None else np.asarray(b, dtype)
    out = sinkhorn_tensorized(a, x, b, y, **kw)
    return float(out) if not isinstance(out, tuple) and np.ndim(out) == 0 else out


def sinkhorn_loss_and_grad(x, y, a=None, b=None, p=2, dtype=np.float64, **kw):
    """Loss and its gradient with respect to x, a (balanced, closed form of SURVEY Appendix A):
    autograd only sees the last soft-mins (sinkhorn_divergence.py:612-623) whose second cloud and
    dual vector are detached, hence  dL/dx = a_i [ dF_ba/dx_i - dF_aa/dx_i ],  dL/da = f_ba - f_aa."""
    x, y = np.asarray(x, dtype), np.asarray(y, dtype)
    a = _uniform(x.shape[-2], x.shape[:-2], dtype) if a is None else np.asarray(a, dtype)
    b = _uniform(y.shape[-2], y.shape[:-2], dtype) if b is None else np.asarray(b, dtype)
    assert kw.get("reach") is None, "closed-form gradient restated for balanced OT only"
    debias = kw.get("debias", True)
    loss, pots, last, (eps, rho, _) = sinkhorn_tensorized(a, x, b, y, p=p, return_internals=True, **kw)
    f_aa, g_bb, g_ab, f_ba = pots
    gx = softmin_points_grad_x(eps, x, y, last["h_ba"], a, p=p, dtype=dtype)
    ga = f_ba.copy()
    if debias:
        gx = gx - softmin_points_grad_x(eps, x, x, last["h_aa"], a, p=p, dtype=dtype)
        ga = ga - f_aa
    return loss, gx, ga


# --------------------------------------------------------------------------------------------------
#  kernel norms  (kernel_samples.py:62-146)
# --------------------------------------------------------------------------------------------------


def kernel_matrix(name, x, y, blur=0.05):
    """gaussian_kernel :62-68, laplacian_kernel :71-77, energy_kernel :80-82 (explicit differences)."""
    if name == "gaussian":
        return np.exp(-squared_distances(x / blur, y / blur) / 2)
    if name == "laplacian":
        return np.exp(-distances(x / blur, y / blur))
    if name == "energy":
        return -distances(x, y)
    raise KeyError(name)


def kernel_conv(name, x, y, v, blur=0.05, row_block=2048):
    """(K_xy @ v)_i, row-blocked."""
    out = np.empty(x.shape[:-1], x.dtype)
    for i0 in range(0, x.shape[-2], row_block):
        K = kernel_matrix(name, x[..., i0:i0 + row_block, :], y, blur)
        out[..., i0:i0 + row_block] = (K * v[..., None, :]).sum(-1)
    return out


def kernel_loss(name, x, y, a=None, b=None, blur=0.05, potentials=False, dtype=np.float64):
    """kernel_loss, kernel_samples.py:92-146."""
    x, y = np.asarray(x, dtype), np.asarray(y, dtype)
    a = _uniform(x.shape[-2], x.shape[:-2], dtype) if a is None else np.asarray(a, dtype)
    b = _uniform(y.shape[-2], y.shape[:-2], dtype) if b is None else np.asarray(b, dtype)
    a_x = kernel_conv(name, x, x, a, blur)
    b_y = kernel_conv(name, y, y, b, blur)
    b_x = kernel_conv(name, x, y, b, blur)
    if potentials:
        a_y = kernel_conv(name, y, x, a, blur)
        return a_x - b_x, b_y - a_y
    out = 0.5 * scal(a, a_x) + 0.5 * scal(b, b_y) - scal(a, b_x)
    return float(out) if np.ndim(out) == 0 else out


def kernel_loss_grad_x(name, x, y, a=None, b=None, blur=0.05, dtype=np.float64):
    """d loss / d x for un-batched clouds (DoubleGrad doubles the symmetric term)."""
    x, y = np.asarray(x, dtype), np.asarray(y, dtype)
    a = _uniform(x.shape[0], (), dtype) if a is None else np.asarray(a, dtype)
    b = _uniform(y.shape[0], (), dtype) if b is None else np.asarray(b, dtype)

    def dk(xa, ya, w):  # sum_j w_j dk/dx(x_i, y_j)
        diff = xa[:, None, :] - ya[None, :, :]
        d2 = (diff * diff).sum(-1)
        if name == "gaussian":
            coef = -np.exp(-d2 / (2 * blur**2)) / blur**2
        else:  # the clamp of utils.py:61 acts on |x/blur - y/blur|^2 (laplacian) or |x - y|^2 (energy)
            live = d2 > (1e-8 * blur**2 if name == "laplacian" else 1e-8)
            d = np.sqrt(np.where(live, d2, 1.0))
            inv = np.where(live, 1.0 / d, 0.0)
            coef = -np.exp(-d / blur) * inv / blur if name == "laplacian" else -inv
        return ((coef * w[None, :])[..., None] * diff).sum(1)

    return a[:, None] * (dk(x, x, a) - dk(x, y, b))


# --------------------------------------------------------------------------------------------------
#  two-scale Sinkhorn  (sinkhorn_samples.py:453-681), dense emulation of the block-sparse reductions
# --------------------------------------------------------------------------------------------------


def grid_cluster(x, size, label_dtype=None):
    """Voxel labels, compacted to 0..C-1 in lexicographic voxel order (pykeops.torch.cluster.grid_cluster
    as used at sinkhorn_samples.py:477).  ``label_dtype``: the precision the bins `(x / size).floor()` are evaluated in — the
    reference does it in the dtype of ``x``, and a float32 cloud of 1e6 points has a handful of coordinates whose quotient falls
    on the other side of an integer in float64; a point in another voxel is another coarse problem (centroids move by 1e-4 of
    a cluster), which showed as a 1e-6 step in dL/dx around that voxel (tools/diag_cfg3_grad.py, round 6)."""
    if label_dtype is not None:
        x, size = np.asarray(x).astype(label_dtype), np.dtype(label_dtype).type(size)
    q = np.floor(x / size).astype(np.int64)
    q -= q.min(0)
    ext = q.max(0) + 1
    code = q[:, 0]
    for d in range(1, q.shape[1]):
        code = code * ext[d] + q[:, d]
    _, lab = np.unique(code, return_inverse=True)
    return lab


def clusterize(a, x, scale, label_dtype=None):
    """sinkhorn_samples.py:453-490: sorted cloud, per-cluster ranges, weighted centroids, summed weights."""
    lab = grid_cluster(x, scale, label_dtype)
    counts = np.bincount(lab)
    a_c = np.bincount(lab, weights=a)
    x_c = np.stack([np.bincount(lab, weights=a * x[:, d]) for d in range(x.shape[1])], 1) / a_c[:, None]
    ends = np.cumsum(counts)
    ranges = np.stack([ends - counts, ends], 1)
    perm = np.argsort(lab, kind="stable")
    return a_c, a[perm], x_c, x[perm], ranges, perm


def _expand_mask(keep, ranges_i, ranges_j, N, M):
    """cluster-level keep mask -> point-level mask on the sorted clouds"""
    li = np.repeat(np.arange(len(ranges_i)), ranges_i[:, 1] - ranges_i[:, 0])
    lj = np.repeat(np.arange(len(ranges_j)), ranges_j[:, 1] - ranges_j[:, 0])
    assert len(li) == N and len(lj) == M
    return keep[li][:, lj]


def softmin_dense_grad_x(eps, C, x, y, h, g, p=2):
    """d/dx of sum_i g_i softmin(eps, C, h)_i for an explicit — possibly +inf-masked — cost matrix C = C(x, y):
    g_i sum_j P_ij dC/dx(x_i, y_j) with P = softmax_j(h_j - C_ij/eps); masked pairs carry P = 0."""
    v = h[None, :] - C / eps
    v = v - v.max(-1, keepdims=True)
    P = np.exp(v)
    P /= P.sum(-1, keepdims=True)
    diff = x[:, None, :] - y[None, :, :]
    if p == 1:
        n2 = (diff * diff).sum(-1, keepdims=True)
        diff = np.where(n2 > 1e-8, diff / np.sqrt(np.where(n2 > 1e-8, n2, 1.0)), 0.0)
    return g[:, None] * (P[..., None] * diff).sum(1)


def sinkhorn_multiscale(a, x, b, y, p=2, blur=0.05, reach=None, diameter=None, scaling=0.5, truncate=5,
                        cluster_scale=None, debias=True, potentials=False, return_info=False, grad=False):
    """Two-scale Sinkhorn on dense matrices.  Cost objects are dicts {C, x, y, ranges_x, ranges_y}; a
    truncated fine object carries C = +inf outside the kept blocks, so softmin_dense ignores those pairs
    exactly as a block-sparse reduction does.

    ``grad`` (balanced loss): also returns dL/dx in the caller's point order.  Autograd only sees the last,
    non-averaged update (sinkhorn_divergence.py:612-623) — or, when the jump falls on the last iteration, the
    extrapolation that replaces it (:533-544, :585-599) — whose first cloud is the fine x and whose second cloud
    and dual vector are detached: dL/dx_i = a_i [d f_ba / dx_i - d f_aa / dx_i] with the (masked) plans of those
    soft-mins; centroids and cluster weights come out of non-differentiable histogram operations."""
    N, D = x.shape
    M = y.shape[0]
    diameter, eps, eps_list, rho = scaling_parameters(x, y, p, blur, reach, diameter, scaling)
    if cluster_scale is None:
        cluster_scale = diameter / (np.sqrt(D) * 2000 ** (1 / D))          # :584-585
    a_c, a, x_c, x, ranges_x, perm_x = clusterize(a, x, cluster_scale)
    b_c, b, y_c, y, ranges_y, perm_y = clusterize(b, y, cluster_scale)

    jumps = [len(eps_list) - 1]                                              # :593-597
    eps_cost = eps
    for i, e in enumerate(eps_list[2:]):
        eps_cost = e                                                         # the reference's shadowed `eps`
        if cluster_scale**p > e:
            jumps = [i + 1]
            break

    def obj(u, v, ru, rv):
        return dict(C=cost_matrix(u, v, p), x=u, y=v, ranges_x=ru, ranges_y=rv)

    def softmin(eps_, Cobj, h):
        return softmin_dense(eps_, Cobj["C"], h)

    def kernel_truncation(C_xy, C_yx, C_xy_f, C_yx_f, f, g, eps_, truncate=None, cost=None):
        if truncate is None:
            return C_xy_f, C_yx_f
        keep = f[:, None] + g[None, :] > C_xy["C"] - truncate * eps_         # :512-514
        mask = _expand_mask(keep, C_xy["ranges_x"], C_xy["ranges_y"], C_xy_f["C"].shape[0], C_xy_f["C"].shape[1])
        info["kept_fraction"].append(float(mask.mean()))
        out_xy = dict(C_xy_f, C=np.where(mask, C_xy_f["C"], np.inf))
        out_yx = dict(C_yx_f, C=np.where(mask.T, C_yx_f["C"], np.inf))
        return out_xy, out_yx

    extrapolations = []

    def extrapolate(f, g, eps_, lam, C_xy, b_log, C_xy_f):                   # :533-544
        extrapolations.append((eps_, C_xy_f["x"], C_xy["y"], b_log + g / eps_))
        return lam * softmin_dense(eps_, cost_matrix(C_xy_f["x"], C_xy["y"], p), b_log + g / eps_)

    info = dict(jumps=jumps, n_clusters=(len(x_c), len(y_c)), cluster_scale=cluster_scale, kept_fraction=[],
                eps_list=eps_list)
    C_xys = [obj(x_c, y_c, ranges_x, ranges_y), obj(x, y, None, None)]
    C_yxs = [obj(y_c, x_c, ranges_y, ranges_x), obj(y, x, None, None)]
    C_xxs = [obj(x_c, x_c, ranges_x, ranges_x), obj(x, x, None, None)] if debias else None
    C_yys = [obj(y_c, y_c, ranges_y, ranges_y), obj(y, y, None, None)] if debias else None

    pots, last = sinkhorn_loop(softmin, [log_weights(a_c), log_weights(a)], [log_weights(b_c), log_weights(b)],
                            C_xxs, C_yys, C_xys, C_yxs, eps_list, rho, jumps=jumps,
                            kernel_truncation=kernel_truncation, truncate=truncate, extrapolate=extrapolate,
                            debias=debias)
    f_aa, g_bb, g_ab, f_ba = pots
    out = sinkhorn_cost(eps_cost, rho, a, b, f_aa, g_bb, g_ab, f_ba, debias=debias, potentials=potentials)
    if potentials:                                                           # :675-679
        F, G = out
        f_x, g_y = np.empty_like(F), np.empty_like(G)
        f_x[perm_x], g_y[perm_y] = F, G
        out = (f_x, g_y)
    if grad:
        assert rho is None and not potentials, "closed-form gradient restated for the balanced loss only"
        if "h_ba" in last:      # the usual case: last update on the (truncated) fine costs
            gs = softmin_dense_grad_x(last["eps"], last["C_xy"]["C"], x, y, last["h_ba"], a, p)
            if debias:
                gs = gs - softmin_dense_grad_x(last["eps"], last["C_xx"]["C"], x, x, last["h_aa"], a, p)
        else:                   # jump on the last iteration: the differentiable step is the extrapolation (fine x vs coarse cloud)
            e, xf, yc, h = extrapolations[0]                      # f_ba: fine x against the coarse y
            gs = softmin_dense_grad_x(e, cost_matrix(xf, yc, p), xf, yc, h, a, p)
            if debias:
                e, xf, xc, h = extrapolations[2]                  # f_aa: fine x against the coarse x
                gs = gs - softmin_dense_grad_x(e, cost_matrix(xf, xc, p), xf, xc, h, a, p)
        gx = np.empty_like(gs)
        gx[perm_x] = gs
        out = (out, gx)
    return (out, info) if return_info else out


def kernel_multiscale(name, a, x, b, y, blur=0.05, truncate=5, diameter=None, cluster_scale=None, potentials=False,
                      grad=False, return_info=False):
    """Block-sparse kernel norm of kernel_samples.py:177-271 on dense masked matrices: clouds centred and rescaled by
    1/blur (:207-211), voxel clusters of the rescaled clouds (:214-226) with alpha / beta-weighted centroids (:229-230), keep-mask
    |c_i - c_j|^2 <= (truncate + cell diameter)^2 on the centroids (:244-252), then kernel_loss (:92-146) restricted to the
    kept blocks.  Returns the loss, or the potentials — like the reference, in cluster-SORTED point order (``perm_x`` / ``perm_y``
    are in ``info``) — optionally with dL/dx in the caller's order (gaussian / laplacian)."""
    a, x, b, y = (np.asarray(t, np.float64) for t in (a, x, b, y))
    if truncate is None or name == "energy":
        out = kernel_loss(name, x, y, a, b, blur=blur, potentials=potentials)
        if grad:
            out = (out, kernel_loss_grad_x(name, x, y, a, b, blur=blur))
        return (out, dict(kept_fraction=[1.0, 1.0, 1.0])) if return_info else out
    N, D = x.shape
    centre = (x.mean(0, keepdims=True) + y.mean(0, keepdims=True)) / 2
    x, y = x - centre, y - centre
    xs, ys = x / blur, y / blur
    if cluster_scale is None:
        diam = max_diameter(xs, ys) if diameter is None else diameter / blur
        cluster_scale = diam / (np.sqrt(D) * 2000 ** (1 / D))
    cell = cluster_scale * np.sqrt(D)
    _, a_s, xc, xs_s, ranges_x, perm_x = clusterize(a, xs, cluster_scale)
    _, b_s, yc, ys_s, ranges_y, perm_y = clusterize(b, ys, cluster_scale)
    x_s, y_s = x[perm_x], y[perm_y]
    reach2 = (truncate + cell) ** 2
    M = y.shape[0]
    m_xx = _expand_mask(squared_distances(xc, xc) <= reach2, ranges_x, ranges_x, N, N)
    m_yy = _expand_mask(squared_distances(yc, yc) <= reach2, ranges_y, ranges_y, M, M)
    m_xy = _expand_mask(squared_distances(xc, yc) <= reach2, ranges_x, ranges_y, N, M)
    K_xx, K_yy, K_xy = (np.where(m, kernel_matrix(name, u, v, blur), 0.0)
                        for m, u, v in ((m_xx, x_s, x_s), (m_yy, y_s, y_s), (m_xy, x_s, y_s)))
    a_x, b_y, b_x = K_xx @ a_s, K_yy @ b_s, K_xy @ b_s
    info = dict(kept_fraction=[float(m.mean()) for m in (m_xx, m_yy, m_xy)], perm_x=perm_x, perm_y=perm_y,
                n_clusters=(len(xc), len(yc)))
    if potentials:
        out = (a_x - b_x, b_y - K_xy.T @ a_s)
    else:
        out = float(0.5 * a_s @ a_x + 0.5 * b_s @ b_y - a_s @ b_x)
        if grad:     # dL/dx_i = a_i [ sum_j a_j dk(x_i,x_j) - sum_j b_j dk(x_i,y_j) ] over the kept pairs (DoubleGrad, :43-54)
            def dk(u, v, w, mask):
                diff = u[:, None, :] - v[None, :, :]
                d2 = (diff * diff).sum(-1)
                if name == "gaussian":
                    coef = -np.exp(-d2 / (2 * blur**2)) / blur**2
                else:
                    live = d2 > 1e-8 * blur**2
                    d = np.sqrt(np.where(live, d2, 1.0))
                    coef = -np.exp(-d / blur) * np.where(live, 1.0 / d, 0.0) / blur
                return ((np.where(mask, coef, 0.0) * w[None, :])[..., None] * diff).sum(1)
            gs = a_s[:, None] * (dk(x_s, x_s, a_s, m_xx) - dk(x_s, y_s, b_s, m_xy))
            gx = np.empty_like(gs)
            gx[perm_x] = gs
            out = (out, gx)
    return (out, info) if return_info else out


# --------------------------------------------------------------------------------------------------
#  measures on regular grids  (_legacy/utils.py:64-279, sinkhorn_images.py, wasserstein_barycenter_images.py)
# --------------------------------------------------------------------------------------------------

def lse_lines(h, eps, p=2):
    """log sum_j exp(h[..., j] - c(i, j)) along the last axis; pixels at i/N, c = (x_i-x_j)^2/(2 eps) or |x_i-x_j|/eps
    (utils.py:242-270: x = arange(N)/N, divided by sqrt(2 eps) (p=2) or eps (p=1))."""
    h = np.asarray(h, np.float64)
    N = h.shape[-1]
    x = np.arange(N, dtype=np.float64) / N
    d = x[:, None] - x[None, :]
    c = d * d / (2.0 * eps) if p == 2 else np.abs(d) / eps
    return logsumexp(h[..., None, :] - c, axis=-1)


def softmin_grid(eps, p, h):
    """Separable soft-min of an image (B,K,N,N) or volume (B,K,N,N,N): one lse_lines pass per spatial axis (utils.py:272-284)."""
    h = np.asarray(h, np.float64)
    last = h.ndim - 1
    out = lse_lines(h, eps, p)
    for axis in range(last - 1, 1, -1):
        out = np.swapaxes(lse_lines(np.swapaxes(out, axis, last), eps, p), axis, last)
    return -eps * out


def grid_subsample(I):
    """Sum over 2^D blocks (utils.py:76-79: 4 * avg_pool2d, 8 * avg_pool3d)."""
    for ax in range(2, I.ndim):
        sh = list(I.shape)
        sh[ax:ax + 1] = [sh[ax] // 2, 2]
        I = I.reshape(sh).sum(ax + 1)
    return I


def grid_pyramid(I):
    """[1x1, ..., NxN] (utils.py:87-96)."""
    levels = [np.asarray(I, np.float64)]
    for _ in range(int(np.log2(I.shape[2]))):
        levels.append(grid_subsample(levels[-1]))
    return levels[::-1]


def grid_upsample(I):
    """2x multilinear interpolation with torch's align_corners=False convention (utils.py:99-101): output sample o reads
    the input at (o + 0.5)/2 - 0.5, clamped at 0, with the right neighbour clamped at the border."""
    for ax in range(2, I.ndim):
        n = I.shape[ax]
        src = np.maximum((np.arange(2 * n) + 0.5) / 2.0 - 0.5, 0.0)
        i0 = np.floor(src).astype(int)
        i1 = np.minimum(i0 + 1, n - 1)
        w = (src - i0).reshape([-1 if k == ax else 1 for k in range(I.ndim)])
        I = (1.0 - w) * np.take(I, i0, axis=ax) + w * np.take(I, i1, axis=ax)
    return I


def log_dens(a):
    """utils.py:104-107."""
    a = np.asarray(a, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.log(a)
    out[a <= 0] = -10000.0
    return out


def sinkhorn_images(a, b, p=2, blur=None, reach=None, scaling=0.5, debias=True, potentials=False):
    """sinkhorn_images.py:26-202 on dense float64 arrays: pyramid, jump schedule, the shared loop, interpolation."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if blur is None:
        blur = 1.0 / a.shape[-1]
    a_s, b_s = grid_pyramid(a)[1:], grid_pyramid(b)[1:]
    a_logs, b_logs = [log_dens(t) for t in a_s], [log_dens(t) for t in b_s]
    C_s = [p] * len(a_logs)
    diameter = 1
    eps = blur**p
    rho = None if reach is None else reach**p
    eps_list = epsilon_schedule(p, diameter, blur, scaling)
    pixel = [diameter / t.shape[-1] for t in a_s]
    current, jumps = pixel.pop(0), []
    for i, e in enumerate(eps_list[1:]):
        if current**p > e:
            jumps.append(i + 1)
            current = pixel.pop(0)
    assert len(jumps) == len(a_s) - 1
    (f_aa, g_bb, g_ab, f_ba), _ = sinkhorn_loop(
        softmin_grid, a_logs, b_logs, C_s, C_s, C_s, C_s, eps_list, rho, jumps=jumps,
        kernel_truncation=lambda C_xy, C_yx, C_xy_f, C_yx_f, f, g, e, truncate=None, cost=None: (C_xy_f, C_yx_f),
        extrapolate=lambda f, g, e, lam, C, b_log, C_f: grid_upsample(f), debias=debias)
    B = a.shape[0]
    flat = lambda t: None if t is None else t.reshape(B, -1)  # noqa: E731
    out = sinkhorn_cost(eps, rho, flat(a), flat(b), flat(f_aa), flat(g_bb), flat(g_ab), flat(f_ba), debias=debias,
                        potentials=potentials)
    if potentials:
        return out[0].reshape(a.shape), out[1].reshape(b.shape)
    return out


def images_barycenter(measures, weights, blur=0, p=2, scaling_N=10, extra_iterations=0):
    """wasserstein_barycenter_images.py:36-93, forward values (``extra_iterations`` = the reference's backward_iterations,
    which also change the returned barycenter)."""
    a_k, w_k = np.asarray(measures, np.float64), np.asarray(weights, np.float64)
    if blur == 0:
        blur = 1.0 / a_k.shape[-1]
    w = w_k[:, :, None, None]

    def iteration(f_k, g_k, d_log, eps, ak_log):
        def bar_of(g):
            return d_log - (softmin_grid(eps, p, ak_log + g / eps) / eps * w).sum(1, keepdims=True)
        bar_log = bar_of(g_k)
        ft_k = softmin_grid(eps, p, ak_log + g_k / eps)
        gt_k = softmin_grid(eps, p, bar_log + f_k / eps)
        f_k, g_k = (f_k + ft_k) / 2, (g_k + gt_k) / 2
        bar_log = bar_of(g_k)
        d_log = 0.5 * (d_log + bar_log + softmin_grid(eps, p, d_log) / eps)
        return f_k, g_k, d_log, bar_log

    ak_log_s = [log_dens(t) for t in grid_pyramid(a_k)[1:]]
    sigma = 1.0
    eps = sigma**p
    f_k, g_k = softmin_grid(eps, p, ak_log_s[0]), softmin_grid(eps, p, ak_log_s[0])
    d_log = np.ones_like(ak_log_s[0]).sum(1, keepdims=True)
    d_log = d_log - logsumexp(d_log.reshape(d_log.shape[0], 1, -1), axis=-1)[..., None, None]
    for n, ak_log in enumerate(ak_log_s):
        for _ in range(scaling_N):
            eps = sigma**p
            f_k, g_k, d_log, bar_log = iteration(f_k, g_k, d_log, eps, ak_log)
            sigma = max(sigma * (2 ** (-1 / scaling_N)), blur)
        if n + 1 < len(ak_log_s):
            f_k, g_k, d_log = grid_upsample(f_k), grid_upsample(g_k), grid_upsample(d_log)
    for _ in range(extra_iterations):
        f_k, g_k, d_log, bar_log = iteration(f_k, g_k, d_log, eps, ak_log)
    return np.exp(bar_log)
