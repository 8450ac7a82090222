"""Grid / image path on the GPU: the HIP line kernel (`glhip_lse_lines_fwd/bwd`) against the NumPy oracle, and the image
Sinkhorn divergence / barycenter built on it against vectors produced by the reference's own code (fp64 and fp32 runs)."""

import ast
import glob
import os

import numpy as np
import pytest
import torch

from geomloss_amd import ImagesBarycenter, hip, sinkhorn_divergence
from geomloss_amd.utils import softmin_grid
from oracle import oracle_np

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
SINKHORN_CASES = sorted(glob.glob(os.path.join(GOLD, "images_p*.npz")) + glob.glob(os.path.join(GOLD, "volumes*.npz")))


@pytest.mark.parametrize("p", [2, 1])
@pytest.mark.parametrize("R,N", [(7, 1), (5, 2), (33, 31), (12, 256), (3, 257), (2, 1000), (1, 4096)])
def test_line_kernel_vs_oracle(cuda, R, N, p):
    rng = np.random.default_rng(R * 1000 + N)
    for eps in (1.0, 1e-2, (1.0 / max(N, 2)) ** p):
        h = (rng.standard_normal((R, N)) * 3).astype(np.float32)
        h[:, ::5] = -10000.0                                   # empty pixels (log_dens)
        if N > 2:
            h[0] = -10000.0
            h[0, N // 2] = 0.0                                 # a single far-away mass: terms ~ -c/eps, no underflow allowed
        ht = torch.from_numpy(h).to(cuda).requires_grad_(True)
        out = hip.lse_lines(ht, eps, p)
        ref = oracle_np.lse_lines(h, eps, p)
        assert np.isfinite(out.detach().cpu().numpy()).all()
        assert np.abs(out.detach().cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max() + 2e-6
        g = rng.standard_normal((R, N)).astype(np.float32)
        (gh,) = torch.autograd.grad(out, [ht], grad_outputs=torch.from_numpy(g).to(cuda))
        x = np.arange(N) / N
        c = (x[:, None] - x[None, :]) ** 2 / (2 * eps) if p == 2 else np.abs(x[:, None] - x[None, :]) / eps
        P = np.exp(h.astype(np.float64)[:, None, :] - c - ref[:, :, None])            # (R, i, j), rows sum to 1
        gref = (g.astype(np.float64)[:, :, None] * P).sum(1)
        # the weights exp(h_j - c_ij - lse_i) are formed from fp32 numbers as large as |lse| (1e4 with empty pixels at
        # -10000): their relative accuracy is a few ulp(|lse|)
        assert np.abs(gh.cpu().numpy() - gref).max() <= (1e-5 + 4 * 1.2e-7 * np.abs(ref).max()) * np.abs(gref).max() + 1e-6


def test_line_kernel_argument_checks(cuda):
    with pytest.raises(NotImplementedError):
        hip.lse_lines(torch.zeros(2, 5000, device=cuda), 0.1, 2)      # longer than the LDS line buffer
    with pytest.raises(NotImplementedError):
        hip.lse_lines(torch.zeros(2, 8, device=cuda), 0.1, 3)
    with pytest.raises(ValueError):
        hip.lse_lines(torch.zeros(2, 8, device=cuda), 0.0, 2)
    assert hip.lse_lines(torch.zeros(0, 8, device=cuda), 0.1, 2).shape == (0, 8)


def test_separable_softmin_vs_reference(cuda):
    g = np.load(os.path.join(GOLD, "images_softmin_grid.npz"))
    for tag in [k[2:] for k in g.files if k.startswith("h_")]:
        h = torch.from_numpy(g["h_" + tag]).float().to(cuda)
        out = softmin_grid(float(g["eps_" + tag]), int(tag[1]), h).cpu().numpy()
        ref = g["out_" + tag]
        assert np.abs(out - ref).max() <= 3e-6 * np.abs(ref).max()


@pytest.mark.parametrize("path", SINKHORN_CASES, ids=[os.path.basename(p)[:-4] for p in SINKHORN_CASES])
def test_image_sinkhorn_vs_reference(cuda, path):
    d = np.load(path)
    kw = ast.literal_eval(str(d["kwargs"]))
    a = torch.from_numpy(d["a"]).float().to(cuda).requires_grad_(True)
    b = torch.from_numpy(d["b"]).float().to(cuda)
    L = sinkhorn_divergence(a, b, **kw)
    (ga,) = torch.autograd.grad(L.sum(), [a])
    F, G = sinkhorn_divergence(a.detach(), b, potentials=True, **kw)
    ref, ref32 = d["loss_f64"], d["loss_f32"]
    # budget: 1e-4 relative (BASELINE); the reference's own fp32 run sits at |ref32 - ref| / |ref| from its fp64 run
    assert np.abs(L.detach().cpu().numpy() - ref).max() <= max(1e-4, 3 * np.abs(ref32 - ref).max() / np.abs(ref).max()) * np.abs(ref).max()
    scale = max(np.abs(d["F_f64"]).max(), np.abs(d["G_f64"]).max())
    assert np.abs(F.cpu().numpy() - d["F_f64"]).max() <= 1e-4 * scale
    assert np.abs(G.cpu().numpy() - d["G_f64"]).max() <= 1e-4 * scale
    assert np.abs(ga.cpu().numpy() - d["ga_f64"]).max() <= 1e-4 * np.abs(d["ga_f64"]).max()


@pytest.mark.parametrize("name", ["barycenter_p2_16", "barycenter_p2_8_blur"])
def test_barycenter_vs_reference(cuda, name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    p, blur, scaling_N = d["cfg"]
    m = torch.from_numpy(d["measures"]).float().to(cuda)
    w = torch.from_numpy(d["weights"]).float().to(cuda).requires_grad_(True)
    target = torch.from_numpy(d["target"]).float().to(cuda)
    bar = ImagesBarycenter(m, w, blur=float(blur), p=int(p), scaling_N=int(scaling_N), backward_iterations=2)
    (gw,) = torch.autograd.grad((bar * target).sum(), [w])
    assert np.abs(bar.detach().cpu().numpy() - d["bar_f64"]).max() <= 2e-4 * np.abs(d["bar_f64"]).max()
    assert np.abs(gw.cpu().numpy() - d["gw_f64"]).max() <= 2e-3 * np.abs(d["gw_f64"]).max()


def test_full_size_images(cuda):
    """256 x 256 images, default blur = one pixel: symmetric, vanishing on identical inputs, finite gradients."""
    g = torch.Generator().manual_seed(0)
    a = torch.rand(2, 1, 256, 256, generator=g).to(cuda) ** 4
    b = torch.rand(2, 1, 256, 256, generator=g).to(cuda) ** 4
    a, b = a / a.sum((2, 3), keepdim=True), b / b.sum((2, 3), keepdim=True)
    a.requires_grad_(True)
    Lab, Lba, Laa = sinkhorn_divergence(a, b), sinkhorn_divergence(b, a.detach()), sinkhorn_divergence(a.detach(), a.detach())
    assert (Lab > 0).all() and (Lab - Lba).abs().max().item() <= 1e-5 * Lab.abs().max().item()
    assert Laa.abs().max().item() <= 1e-6
    (ga,) = torch.autograd.grad(Lab.sum(), [a])
    assert torch.isfinite(ga).all()


def test_image_loop_replayed_from_a_hipgraph(cuda):
    """Graph mode: the data-independent annealing loop is captured once per shape and replayed; same values as eager."""
    from geomloss_amd import sinkhorn_samples as ss

    g = torch.Generator().manual_seed(3)
    res = {}
    for mode in (False, True, True):           # eager, capture, replay (on new data)
        ss.set_graph_mode(mode)
        try:
            out = []
            for _ in range(2):
                a = torch.rand(2, 1, 32, 32, generator=g).to(cuda) ** 2
                b = torch.rand(2, 1, 32, 32, generator=g).to(cuda) ** 2
                a, b = a / a.sum((2, 3), keepdim=True), b / b.sum((2, 3), keepdim=True)
                a.requires_grad_(True)
                L = sinkhorn_divergence(a, b, p=2)
                (ga,) = torch.autograd.grad(L.sum(), [a])
                out.append((L.detach(), ga))
            res.setdefault(mode, []).append(out)
        finally:
            ss.set_graph_mode(False)
        g.manual_seed(3)
    eager = res[False][0]
    for run in res[True]:
        for (L0, g0), (L1, g1) in zip(eager, run):
            assert (L0 - L1).abs().max().item() <= 1e-7 * L0.abs().max().item()
            assert (g0 - g1).abs().max().item() <= 1e-6 * g0.abs().max().item()


def test_hard_c_transform_on_grids(cuda):
    """utils.C_transform (glhip_max_lines_fwd, one pass per axis) against the dense definition
    F(x_i) = max_j [G(x_j) - |x_i - x_j|^2 / (2 tau)] on 1-D, 2-D and 3-D grids (_legacy/utils.py:116-182)."""
    from geomloss_amd.utils import C_transform

    g = torch.Generator().manual_seed(0)
    for shape, tau in (((3, 37), 1.0), ((2, 24, 24), 2.5), ((2, 9, 9, 9), 0.7)):
        G = torch.randn(shape, generator=g).to(cuda) * 3
        out = C_transform(G, tau=tau, p=2).double().cpu()
        D, N = len(shape) - 1, shape[1]
        grid = torch.stack(torch.meshgrid(*([torch.arange(N, dtype=torch.float64)] * D), indexing="ij"), -1).reshape(-1, D)
        C = ((grid[:, None, :] - grid[None, :, :]) ** 2).sum(-1) / (2 * tau)
        ref = (G.double().cpu().reshape(shape[0], 1, -1) - C[None]).max(-1)[0].reshape(shape)
        assert (out - ref).abs().max() < 1e-5
    with pytest.raises(NotImplementedError):
        C_transform(torch.zeros(1, 8, device=cuda), p=1)
