"""Generates tests/golden/*.npz by running the REFERENCE implementation (jeanfeydy/geomloss 0.3.1,
mounted read-only at /root/reference) on seeded inputs.  Run once in the build container:

    python tests/golden/make_golden.py

The reference's own tests never touch ``SamplesLoss``; these vectors are what pins our oracle and our
kernels to the reference's tensorized backend.  Only this script reads /root/reference; the tests read
the committed .npz files.
"""

import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/src")
import geomloss  # noqa: E402  (the reference)
from geomloss._legacy.sinkhorn_samples import softmin_tensorized, cost_routines  # noqa: E402

assert geomloss.__version__ == "0.3.1"
OUT = os.path.dirname(os.path.abspath(__file__))


def clouds(seed, B, N, M, D, weights):
    g = torch.Generator().manual_seed(seed)
    shape_x = (N, D) if B == 0 else (B, N, D)
    shape_y = (M, D) if B == 0 else (B, M, D)
    x = torch.rand(shape_x, generator=g, dtype=torch.float64)
    y = torch.rand(shape_y, generator=g, dtype=torch.float64) * 0.9 + 0.05
    if weights:
        a = torch.rand(shape_x[:-1], generator=g, dtype=torch.float64) + 0.1
        b = torch.rand(shape_y[:-1], generator=g, dtype=torch.float64) + 0.1
        a, b = a / a.sum(-1, keepdim=True), b / b.sum(-1, keepdim=True)
    else:
        a = torch.full(shape_x[:-1], 1.0 / N, dtype=torch.float64)
        b = torch.full(shape_y[:-1], 1.0 / M, dtype=torch.float64)
    return a, x, b, y


def run(kwargs, a, x, b, y, dtype):
    a, x, b, y = (t.to(dtype) for t in (a, x, b, y))
    x = x.clone().requires_grad_(True)
    a = a.clone().requires_grad_(True)
    L = geomloss.SamplesLoss(backend="tensorized", **kwargs)(a, x, b, y)
    gx, ga = torch.autograd.grad(L.sum(), [x, a])
    F, G = geomloss.SamplesLoss(backend="tensorized", potentials=True, **kwargs)(a.detach(), x.detach(), b, y)
    return dict(loss=L.detach().numpy(), gx=gx.numpy(), ga=ga.numpy(), F=F.detach().numpy(), G=G.detach().numpy())


CASES = {
    # name: (SamplesLoss kwargs, seed, B, N, M, D, random weights)
    "sinkhorn_p2_d2": (dict(loss="sinkhorn", p=2, blur=0.05), 1, 0, 230, 190, 2, False),
    "sinkhorn_p2_d3_w": (dict(loss="sinkhorn", p=2, blur=0.05), 2, 0, 210, 260, 3, True),
    "sinkhorn_p2_d1": (dict(loss="sinkhorn", p=2, blur=0.05), 3, 0, 150, 170, 1, True),
    "sinkhorn_p1_d3": (dict(loss="sinkhorn", p=1, blur=0.05), 4, 0, 200, 180, 3, True),
    "sinkhorn_p2_nodebias": (dict(loss="sinkhorn", p=2, blur=0.1, debias=False), 5, 0, 160, 200, 3, True),
    "sinkhorn_p2_reach": (dict(loss="sinkhorn", p=2, blur=0.05, reach=0.3), 6, 0, 180, 220, 2, True),
    "sinkhorn_p2_batch": (dict(loss="sinkhorn", p=2, blur=0.05, diameter=1.8), 7, 3, 120, 140, 3, True),
    "sinkhorn_p2_d5": (dict(loss="sinkhorn", p=2, blur=0.1), 8, 0, 130, 110, 5, True),
    "sinkhorn_p2_scaling9": (dict(loss="sinkhorn", p=2, blur=0.05, scaling=0.9), 9, 0, 150, 150, 3, False),
    "sinkhorn_p2_big": (dict(loss="sinkhorn", p=2, blur=0.05), 10, 0, 1500, 1300, 3, True),
    "gaussian_d3": (dict(loss="gaussian", blur=0.1), 11, 0, 220, 240, 3, True),
    "laplacian_d2": (dict(loss="laplacian", blur=0.1), 12, 0, 200, 210, 2, True),
    "energy_d3": (dict(loss="energy"), 13, 0, 190, 230, 3, True),
    "gaussian_batch": (dict(loss="gaussian", blur=0.2), 14, 2, 100, 120, 3, True),
    "gaussian_d6": (dict(loss="gaussian", blur=0.3), 15, 0, 90, 100, 6, True),
    # round 3: the D = 4 ... 16 matrix-core kernels, the distance kernels with weights / in a batch, p = 1 with a reach
    "sinkhorn_p2_d8": (dict(loss="sinkhorn", p=2, blur=0.2), 18, 0, 120, 100, 8, True),
    "sinkhorn_p2_d16": (dict(loss="sinkhorn", p=2, blur=0.3), 19, 0, 110, 130, 16, True),
    "gaussian_d12": (dict(loss="gaussian", blur=0.5), 20, 0, 100, 90, 12, True),
    "laplacian_d3": (dict(loss="laplacian", blur=0.1), 21, 0, 210, 190, 3, True),
    "energy_d2": (dict(loss="energy"), 22, 0, 180, 200, 2, True),
    "laplacian_batch": (dict(loss="laplacian", blur=0.2), 23, 2, 110, 100, 3, True),
    "sinkhorn_p1_reach_d2": (dict(loss="sinkhorn", p=1, blur=0.05, reach=0.5), 24, 0, 170, 190, 2, True),
    "sinkhorn_p1_batch": (dict(loss="sinkhorn", p=1, blur=0.05, diameter=1.8), 25, 2, 100, 120, 3, True),
    # round 5: the distance reductions (p = 1, laplacian, energy) for 4 <= D <= 16 on the matrix cores (csrc/glhip_dist_xd.h)
    "sinkhorn_p1_d5": (dict(loss="sinkhorn", p=1, blur=0.1), 26, 0, 130, 150, 5, True),
    "laplacian_d8": (dict(loss="laplacian", blur=0.3), 27, 0, 140, 120, 8, True),
    "energy_d6": (dict(loss="energy"), 28, 0, 150, 130, 6, True),
    "sinkhorn_p1_d12_batch": (dict(loss="sinkhorn", p=1, blur=0.2, diameter=3.0), 29, 2, 90, 100, 12, True),
    # mid-size cases: what pins the chunked full-size oracle (oracle/oracle_torch64.py) beyond the sizes NumPy handles
    "sinkhorn_p2_n8000": (dict(loss="sinkhorn", p=2, blur=0.05), 16, 0, 8000, 7000, 3, True),
    "gaussian_n8000": (dict(loss="gaussian", blur=0.05), 17, 0, 8000, 7000, 3, True),
}


def main():
    torch.set_num_threads(4)
    only = set(sys.argv[1:])          # `python make_golden.py name ...` regenerates just those cases
    for name, (kw, seed, B, N, M, D, wts) in CASES.items():
        if only and name not in only:
            continue
        a, x, b, y = clouds(seed, B, N, M, D, wts)
        rec = dict(a=a.numpy(), x=x.numpy(), b=b.numpy(), y=y.numpy(), kwargs=repr(kw))
        for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
            for k, v in run(kw, a, x, b, y, dt).items():
                rec[f"{k}_{tag}"] = v
        # inputs are stored in float32 (what the kernels see) when that loses nothing for the f32 run
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, "loss f64", rec["loss_f64"], "f32", rec["loss_f32"])
    if only:
        return

    # BASELINE config 1 exactly: SamplesLoss('sinkhorn', p=2, blur=.05) tensorized, N=M=2000, 2D, fp32, CPU
    torch.manual_seed(0)
    x, y = torch.rand(2000, 2), torch.rand(2000, 2)
    L32 = geomloss.SamplesLoss("sinkhorn", p=2, blur=0.05, backend="tensorized")(x, y).item()
    L64 = geomloss.SamplesLoss("sinkhorn", p=2, blur=0.05, backend="tensorized")(x.double(), y.double()).item()
    F, G = geomloss.SamplesLoss("sinkhorn", p=2, blur=0.05, backend="tensorized", potentials=True)(x.double(), y.double())
    np.savez_compressed(os.path.join(OUT, "cfg1_n2000_d2.npz"), x=x.numpy(), y=y.numpy(), loss_f32=L32, loss_f64=L64,
                        F_f64=F.numpy().astype(np.float32), G_f64=G.numpy().astype(np.float32))
    print("cfg1", L32, L64)

    # the soft-min itself (sinkhorn_samples.py:32-71) on one explicit cost matrix
    g = torch.Generator().manual_seed(21)
    xs = torch.rand(2, 300, 3, generator=g, dtype=torch.float64)
    ys = torch.rand(2, 280, 3, generator=g, dtype=torch.float64)
    h = torch.randn(2, 280, generator=g, dtype=torch.float64)
    rec = dict(x=xs.numpy(), y=ys.numpy(), h=h.numpy())
    for p in (1, 2):
        for eps in (1.0, 0.05**p):
            C = cost_routines[p](xs, ys)
            rec[f"softmin_p{p}_eps{eps:g}"] = softmin_tensorized(eps, C, h).numpy()
    np.savez_compressed(os.path.join(OUT, "softmin_tensorized.npz"), **rec)
    print("softmin vectors written")


if __name__ == "__main__":
    main()
