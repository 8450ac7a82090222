"""Grid / image path without a GPU: the NumPy oracle against vectors produced by the reference's own
`sinkhorn_images.py` / `wasserstein_barycenter_images.py` (tests/golden/make_golden_images.py), and the host-side helpers."""

import ast
import glob
import os

import numpy as np
import pytest
import torch

from oracle import oracle_np

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SINKHORN_CASES = sorted(glob.glob(os.path.join(GOLD, "images_p*.npz")) + glob.glob(os.path.join(GOLD, "volumes*.npz")))


def test_oracle_separable_softmin_matches_the_reference():
    g = np.load(os.path.join(GOLD, "images_softmin_grid.npz"))
    tags = [k[2:] for k in g.files if k.startswith("h_")]
    assert len(tags) == 3
    for tag in tags:
        out = oracle_np.softmin_grid(float(g["eps_" + tag]), int(tag[1]), g["h_" + tag])
        assert np.abs(out - g["out_" + tag]).max() < 1e-12


@pytest.mark.parametrize("path", SINKHORN_CASES, ids=[os.path.basename(p)[:-4] for p in SINKHORN_CASES])
def test_oracle_image_sinkhorn_matches_the_reference(path):
    d = np.load(path)
    kw = ast.literal_eval(str(d["kwargs"]))
    loss = oracle_np.sinkhorn_images(d["a"], d["b"], **kw)
    F, G = oracle_np.sinkhorn_images(d["a"], d["b"], potentials=True, **kw)
    assert np.abs(loss - d["loss_f64"]).max() <= 1e-10 * np.abs(d["loss_f64"]).max()
    assert np.abs(F - d["F_f64"]).max() < 1e-10 and np.abs(G - d["G_f64"]).max() < 1e-10


@pytest.mark.parametrize("name", ["barycenter_p2_16", "barycenter_p2_8_blur"])
def test_oracle_barycenter_matches_the_reference(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    p, blur, scaling_N = d["cfg"]
    bar = oracle_np.images_barycenter(d["measures"], d["weights"], blur=float(blur), p=int(p), scaling_N=int(scaling_N),
                                      extra_iterations=2)
    assert np.abs(bar - d["bar_f64"]).max() <= 1e-10 * np.abs(d["bar_f64"]).max()


def test_grid_helpers_agree_with_the_oracle_restatement():
    from geomloss_amd import utils

    g = torch.Generator().manual_seed(0)
    for shape in ((2, 3, 8, 8), (1, 2, 4, 4, 4)):
        I = torch.rand(shape, generator=g, dtype=torch.float64)
        levels, ref = utils.pyramid(I), oracle_np.grid_pyramid(I.numpy())
        assert [tuple(t.shape) for t in levels] == [t.shape for t in ref]
        assert all(np.abs(t.numpy() - r).max() < 1e-12 for t, r in zip(levels, ref))
        assert np.abs(utils.upsample(I).numpy() - oracle_np.grid_upsample(I.numpy())).max() < 1e-12
        I[0, 0, 0] = 0
        assert np.array_equal(utils.log_dens(I).numpy(), oracle_np.log_dens(I.numpy()))
    assert utils.dimension(torch.zeros(1, 1, 4, 4)) == 2 and utils.dimension(torch.zeros(1, 1, 4, 4, 4)) == 3


def test_image_path_is_exported_like_the_reference_and_needs_a_gpu():
    import geomloss_amd
    from geomloss_amd import ImagesBarycenter, sinkhorn_divergence

    assert callable(sinkhorn_divergence) and callable(ImagesBarycenter)
    assert hasattr(geomloss_amd.sinkhorn_divergence, "sinkhorn_loop")     # ... and it is still the solver module
    a = torch.rand(1, 1, 4, 4)
    with pytest.raises(RuntimeError, match="GPU"):                         # no CPU fallback for the kernels
        sinkhorn_divergence(a / a.sum(), a / a.sum())
    with pytest.raises(ValueError, match="too small"):
        sinkhorn_divergence(a, a, scaling=0.3)
    with pytest.raises(NotImplementedError):
        sinkhorn_divergence(a, a, cost=lambda x, y: x)
