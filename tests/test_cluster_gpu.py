"""The device-side cluster pyramid (``glhip_grid_cluster``, ``glhip_block_ranges``).

Two references.  ``geomloss_amd/cluster.py`` is the package's OWN torch restatement of the pykeops helpers (the fallback for user
labels / D > 3): comparing the kernels with it checks that the two product paths agree, not that either is right.  The independent
check is ``oracle_np.grid_cluster`` / ``clusterize`` (test infrastructure; tests/test_host_logic.py ties it to the same semantics
on the CPU): ``test_grid_cluster_c_abi_vs_numpy_oracle`` compares the raw C-ABI call with it, and the end-to-end multiscale losses
of tests/test_samples_loss_gpu.py / test_full_size_gpu.py are computed against oracles built on it."""

import numpy as np
import pytest
import torch

from geomloss_amd import SamplesLoss, cluster, hip
from geomloss_amd.utils import squared_distances
from oracle import oracle_np

pytestmark = pytest.mark.gpu


def _cloud(seed, N, D, dev, kind="cube"):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, D, generator=g)
    if kind == "sphere" and D == 3:
        x = torch.randn(N, D, generator=g)
        x[:, 0] += 1
        x = x / (2 * x.norm(dim=1, keepdim=True))
    w = torch.rand(N, generator=g) + 0.1
    return x.to(dev), (w / w.sum()).to(dev)


@pytest.mark.parametrize("N,D,kind,scale,pre_div", [(5000, 3, "cube", 0.13, 1.0), (200_000, 3, "sphere", 0.05, 1.0),
                                                    (3000, 2, "cube", 0.07, 1.0), (1000, 1, "cube", 0.011, 1.0),
                                                    (20_000, 3, "cube", 1.7, 0.05), (1, 3, "cube", 0.5, 1.0)])
def test_grid_cluster_matches_the_torch_restatement(cuda, N, D, kind, scale, pre_div):
    x, w = _cloud(N + D, N, D, cuda, kind)
    a_c, a_s, x_c, x_s, ranges, perm = cluster.clusterize_device(w, x, scale, pre_div=pre_div)
    # torch restatement: labels of x / pre_div, stable sort, fp64 segment sums
    xs = x / pre_div
    lab = cluster.grid_cluster(xs, scale)
    r_ref, c_ref, w_ref = cluster.cluster_ranges_centroids(xs, lab, weights=w)
    perm_ref = torch.sort(lab.view(-1), stable=True)[1]
    assert torch.equal(perm, perm_ref)                      # same clusters, same (stable) order inside each cluster
    assert torch.equal(ranges, r_ref)
    assert torch.equal(x_s, x[perm_ref]) and torch.equal(a_s, w[perm_ref])
    assert (a_c - w_ref).abs().max() <= 1e-6 * w_ref.abs().max()
    assert (x_c - c_ref).abs().max() <= 2e-6 * max(1.0, c_ref.abs().max().item())
    # ... and the NumPy oracle of the same helpers
    a_o, _, xc_o, _, r_o, perm_o = oracle_np.clusterize(w.double().cpu().numpy(), xs.double().cpu().numpy(), scale)
    assert np.array_equal(r_o, ranges.cpu().numpy()) and np.array_equal(perm_o, perm.cpu().numpy())
    assert np.abs(xc_o - x_c.cpu().numpy()).max() < 2e-6 * max(1.0, np.abs(xc_o).max())
    # determinism: bit-identical on a second call
    again = cluster.clusterize_device(w, x, scale, pre_div=pre_div)
    assert torch.equal(again[0], a_c) and torch.equal(again[2], x_c)


@pytest.mark.parametrize("N,D,scale,weighted", [(30_000, 3, 0.09, True), (7000, 2, 0.04, False), (513, 1, 0.003, True)])
def test_grid_cluster_c_abi_vs_numpy_oracle(cuda, N, D, scale, weighted):
    """``glhip_grid_cluster`` through its raw binding against ``oracle_np.clusterize`` (sinkhorn_samples.py:453-490 restated in
    NumPy, no code shared with the package): permutation, ranges, sorted cloud and weights bit for bit; centroids / cluster
    weights to fp32 rounding of the float64 sums."""
    x, w = _cloud(100 + N, N, D, cuda)
    perm, x_s, w_s, ranges, cents, w_c = hip.grid_cluster_raw(x.contiguous(), w if weighted else None, scale)
    xn = x.double().cpu().numpy()
    wn = w.double().cpu().numpy() if weighted else np.ones(N)
    a_c, a_s, x_c, xs_o, r_o, perm_o = oracle_np.clusterize(wn, xn, scale)
    assert np.array_equal(perm.cpu().numpy(), perm_o) and np.array_equal(ranges.cpu().numpy(), r_o)
    assert np.array_equal(x_s.double().cpu().numpy(), xs_o)
    assert np.array_equal(w_s.double().cpu().numpy(), a_s)
    assert np.abs(w_c.double().cpu().numpy() - a_c).max() <= 1e-6 * a_c.max()
    assert np.abs(cents.double().cpu().numpy() - x_c).max() <= 2e-6


def test_grid_cluster_bf16_gradients_and_errors(cuda):
    x, w = _cloud(3, 4000, 3, cuda)
    xb = x.bfloat16()
    a_c, a_s, x_c, x_s, ranges, perm = cluster.clusterize_device(w, xb, 0.2)
    lab = cluster.grid_cluster(xb.float(), 0.2)
    assert torch.equal(perm, torch.sort(lab.view(-1), stable=True)[1]) and x_s.dtype == torch.bfloat16 and a_c.dtype == torch.float32
    # the sorted cloud / weights stay attached to the autograd graph of x / a
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    _, a_s, _, x_s, _, perm = cluster.clusterize_device(wg, xg, 0.2)
    gx, gw = torch.autograd.grad((x_s * torch.arange(4000.0, device=cuda)[:, None]).sum() + (a_s * 2).sum(), [xg, wg])
    expect = torch.empty(4000, device=cuda)
    expect[perm] = torch.arange(4000.0, device=cuda)
    assert torch.equal(gx[:, 0], expect) and torch.equal(gw, torch.full_like(gw, 2.0))
    with pytest.raises(ValueError, match="2\\^21"):
        cluster.clusterize_device(w, x * 1e6, 1e-3)
    lib = hip.load_library()
    assert lib.glhip_grid_cluster(None, None, 10, 4, 0, 1.0, 1.0, None, None, None, None, None, None, None, None, 0, None) == -2


def _mask_of(rg, Ci, Cj):
    """Cluster-level boolean mask described by a BlockRanges (row orientation)."""
    starts = {int(s): k for k, s in enumerate(rg.ranges_j[:, 0].tolist())}
    ends = {int(e): k for k, e in enumerate(rg.ranges_j[:, 1].tolist())}
    sl, red = rg.slices_i.tolist(), rg.redranges_j.tolist()
    mask = np.zeros((Ci, Cj), bool)
    for i in range(Ci):
        for q in range(sl[i - 1] if i else 0, sl[i]):
            mask[i, starts[red[q][0]]:ends[red[q][1]] + 1] = True
    return mask


@pytest.mark.parametrize("kind", ["within", "dual_slack"])
@pytest.mark.parametrize("Ci,Cj,exact", [(70, 90, False), (257, 130, False), (1, 5, False), (257, 130, True), (3000, 2500, True)])
def test_block_ranges_match_from_matrix(cuda, kind, Ci, Cj, exact, monkeypatch):
    """``exact``: interval buffers sized from the counting pass (``glhip_block_ranges_count``: what big cluster grids get)
    instead of for the worst case."""
    if exact:
        monkeypatch.setattr(hip, "_RANGES_WORST_CASE_MAX", 0)
    g = torch.Generator().manual_seed(Ci + Cj)
    xc, yc = torch.rand(Ci, 3, generator=g).to(cuda), torch.rand(Cj, 3, generator=g).to(cuda)
    # row ranges with a few gaps, so that not every pair of neighbouring clusters is adjacent in memory
    def ranges(C, seed):
        sizes = torch.randint(1, 9, (C,), generator=torch.Generator().manual_seed(seed))
        gaps = (torch.rand(C, generator=torch.Generator().manual_seed(seed + 1)) < 0.2).long()
        start = torch.cumsum(sizes + gaps, 0) - sizes
        return torch.stack((start, start + sizes), 1).int().to(cuda)
    ri, rj = ranges(Ci, 1), ranges(Cj, 2)
    if kind == "within":
        thr = 0.25
        lhs = squared_distances(xc.double(), yc.double())
        keep = lhs <= thr
        margin = (lhs - thr).abs()
        rg = cluster.block_ranges_device("within", xc, yc, None, None, ri, rj, thr)
    else:
        f, gg = torch.randn(Ci, generator=g).to(cuda) * 0.1, torch.randn(Cj, generator=g).to(cuda) * 0.1
        thr = 5 * 0.02
        C = squared_distances(xc.double(), yc.double()) / 2
        lhs = f.double()[:, None] + gg.double()[None, :] - C + thr
        keep = lhs > 0
        margin = lhs.abs()
        rg = cluster.block_ranges_device("dual_slack", xc, yc, f, gg, ri, rj, thr, p=2)
    ref = cluster.from_matrix(ri, rj, keep)
    got, want = _mask_of(rg, Ci, Cj), keep.cpu().numpy()
    diff = got != want
    assert diff.sum() <= 2 + Ci * Cj // 100_000 and (margin.cpu().numpy()[diff] < 1e-6).all()     # only fp32-borderline pairs may differ
    if exact:
        assert rg.redranges_j.shape[0] == max(int(rg.slices_i[-1]), int(rg.slices_j[-1]), 1)
    got_t, want_t = _mask_of(rg.t(), Cj, Ci), want.T
    assert ((got_t != want_t) == diff.T).all()
    if diff.sum() == 0:   # identical CSR structure, interval for interval (merged adjacent clusters included)
        for a_, b_ in ((rg.slices_i, ref.slices_i), (rg.slices_j, ref.slices_j)):
            assert torch.equal(a_, b_)
        n, nt = int(ref.slices_i[-1]), int(ref.slices_j[-1])
        assert torch.equal(rg.redranges_j[:n], ref.redranges_j) and torch.equal(rg.redranges_i[:nt], ref.redranges_i)


@pytest.mark.parametrize("C,exact", [(300, False), (2197, False), (900, True)])
def test_symmetric_patterns_build_one_orientation(cuda, C, exact, monkeypatch):
    """The debiasing terms of a Sinkhorn divergence apply the keep rule to (x, x) with one potential (sinkhorn_divergence.py:284-289):
    ``glhip_block_ranges`` with slices_cols = red_rows = NULL builds the row-major pattern only and the host uses it for both
    orientations.  Same CSR arrays as the two-orientation call, interval for interval, and the same mask as ``from_matrix``."""
    if exact:
        monkeypatch.setattr(hip, "_RANGES_WORST_CASE_MAX", 0)
    g = torch.Generator().manual_seed(C)
    xc = torch.rand(C, 3, generator=g).to(cuda)
    f = (torch.randn(C, generator=g) * 0.05).to(cuda)
    sizes = torch.randint(1, 9, (C,), generator=g)
    start = torch.cumsum(sizes, 0) - sizes
    ri = torch.stack((start, start + sizes), 1).int().to(cuda)
    thr = 5 * 0.01
    both = cluster.block_ranges_device("dual_slack", xc, xc, f, f, ri, ri, thr, p=2)
    one = cluster.block_ranges_device("dual_slack", xc, xc, f, f, ri, ri, thr, p=2, symmetric=True)
    n = int(both.slices_i[-1])
    assert torch.equal(one.slices_i, both.slices_i) and torch.equal(one.redranges_j[:n], both.redranges_j[:n])
    assert torch.equal(one.slices_j, both.slices_j) and torch.equal(one.redranges_i[:n], both.redranges_i[:n])      # the pattern IS symmetric
    assert one.slices_j is one.slices_i and one.redranges_i is one.redranges_j
    m = _mask_of(one, C, C)
    assert (m == m.T).all() and (_mask_of(one.t(), C, C) == m).all()


def test_multiscale_losses_are_the_same_on_both_clustering_paths(cuda, monkeypatch):
    """End to end: device clustering + device keep rule vs the torch helpers (forced by disabling the fast path)."""
    g = torch.Generator().manual_seed(5)
    x, y = torch.rand(6000, 3, generator=g).to(cuda), (torch.rand(7000, 3, generator=g) * 0.7 + 0.2).to(cuda)
    losses = {}
    for native in (True, False):
        if not native:
            monkeypatch.setattr(cluster, "native_clustering_applies", lambda *a, **k: False)
            import geomloss_amd.kernel_samples as ks
            import geomloss_amd.sinkhorn_samples as ss
            monkeypatch.setattr(ks, "native_clustering_applies", lambda *a, **k: False)
            monkeypatch.setattr(ss, "native_clustering_applies", lambda *a, **k: False)
            monkeypatch.setattr(ss, "native_keep_rule_applies", lambda *a, **k: False)      # (the keep rule has its own predicate since round 6)
        xg = x.clone().requires_grad_(True)
        Ls = SamplesLoss("sinkhorn", p=2, blur=0.05, scaling=0.7, backend="multiscale")(xg, y)
        (gs,) = torch.autograd.grad(Ls, [xg])
        Lk = SamplesLoss("gaussian", blur=0.05, truncate=3, backend="multiscale")(xg, y)
        (gk,) = torch.autograd.grad(Lk, [xg])
        F, G = SamplesLoss("sinkhorn", p=2, blur=0.05, scaling=0.7, backend="multiscale", potentials=True)(x, y)
        losses[native] = (Ls.item(), gs, Lk.item(), gk, F, G)
    a, b = losses[True], losses[False]
    assert abs(a[0] - b[0]) <= 2e-6 * abs(b[0]) and abs(a[2] - b[2]) <= 2e-6 * abs(b[2])
    assert (a[1] - b[1]).abs().max() <= 1e-5 * b[1].abs().max() and (a[3] - b[3]).abs().max() <= 1e-5 * b[3].abs().max()
    assert (a[4] - b[4]).abs().max() <= 1e-6 and (a[5] - b[5]).abs().max() <= 1e-6


def test_multiscale_loss_is_bitwise_reproducible(cuda):
    """Same inputs, same bits: no floating-point atomics anywhere between the clouds and the loss (deterministic centroids,
    fixed-order block-sparse reductions)."""
    g = torch.Generator().manual_seed(9)
    x, y = torch.rand(150_000, 3, generator=g).to(cuda), torch.rand(150_000, 3, generator=g).to(cuda)
    L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale")
    vals = []
    for _ in range(3):
        xg = x.clone().requires_grad_(True)
        v = L(xg, y)
        (gx,) = torch.autograd.grad(v, [xg])
        vals.append((v.item(), gx))
    assert vals[0][0] == vals[1][0] == vals[2][0]
    assert torch.equal(vals[0][1], vals[1][1]) and torch.equal(vals[0][1], vals[2][1])
