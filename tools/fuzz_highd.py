"""Randomised cross-check of the 4 <= D <= 16 matrix-core kernels against the direct-difference VALU kernels (GPU).

    python tools/fuzz_highd.py [cases=200] [seed=0]

The companion of tools/fuzz_kernels.py (D <= 3) for the round-4/5 families: the many-coordinate exponent kernels (p = 2, both
K layouts), the distance kernels (p = 1 soft-min, laplacian and energy products) and their gradients, and the fused four-softmin
iteration in both exponents.  Shapes straddle the column tile, the 32-column group and the row-block boundaries; clouds hold
duplicated points (distance 0: the near-pair path of the distance kernels); dual vectors hold -inf (massless columns).
Prints the worst normalised error per code path; exits non-zero when one exceeds the tolerance of tests/test_xd_kernels_gpu.py.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from geomloss_amd import hip


def main(n_cases=200, seed=0):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    worst = {}
    edges = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1025, 2049]

    for case in range(n_cases):
        pick = lambda: int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 5000))  # noqa: E731
        N, M, D = pick(), pick(), int(rng.integers(4, 17))
        B = None if rng.random() < 0.7 else int(rng.integers(2, 5))
        eps = float(10 ** rng.uniform(-3, 1))
        scale = float(10 ** rng.uniform(-1, 1))
        shp = (lambda n: (n, D)) if B is None else (lambda n: (B, n, D))
        xn, yn = rng.random(shp(N)) * scale, rng.random(shp(M)) * scale
        if rng.random() < 0.4:          # coincident points: |x_i - y_j| = 0 exactly for some pairs
            k = min(N, M, 40)
            yn[..., :k, :] = xn[..., :k, :]
        x = torch.tensor(xn, dtype=torch.float32, device=dev)
        y = torch.tensor(yn, dtype=torch.float32, device=dev)
        hn = rng.standard_normal(shp(M)[:-1]) * rng.choice([0.1, 3.0, 50.0])
        if rng.random() < 0.3:
            hn[..., rng.integers(0, M)] = -np.inf if M > 1 else -100.0
        if rng.random() < 0.2:
            hn = hn + np.arange(M) * rng.choice([0.5, -0.5])
        h = torch.tensor(hn, dtype=torch.float32, device=dev)
        if rng.random() < 0.25:
            x, y = x.bfloat16(), y.bfloat16()
        diam2 = D * scale * scale
        diam = diam2 ** 0.5
        cfg = dict(N=N, M=M, D=D, B=B, eps=eps, scale=scale, dtype=str(x.dtype).replace("torch.", ""))

        def note(name, err, **extra):
            if not (err <= worst.get(name, (0,))[0]):
                worst[name] = (err, dict(cfg, **extra))

        # ---- soft-min forward, p = 2 (exponent kernels) and p = 1 (distance kernels) ----
        for p in (2, 1):
            ref = hip.softmin(eps, x, y, h, p=p, flags=hip.FLAG_DIRECT)
            cost = diam2 if p == 2 else diam
            tol = 4e-7 * cost + 2e-6 * ref.abs().max().item() + 1e-30
            if p == 1:    # the squared distance is a chain of products: error ~ 2^-22 diam^2 under a root, absolute sqrt at 0
                tol += 1e-3 * diam * 2e-3
            variants = [(f"p{p} softmin", 0), (f"p{p} softmin nosplit", hip.FLAG_NO_SPLIT)]
            if p == 2 and diam2 / eps < 1e5:
                variants += [("p2 softmin f16x2", hip.FLAG_F16X2), ("p2 softmin f16x2 nosplit", hip.FLAG_F16X2 | hip.FLAG_NO_SPLIT)]
            for name, flags in variants:
                out = hip.softmin(eps, x, y, h, p=p, flags=flags)
                t = tol + (2 * 2.4e-7 * eps * 0.6931 if flags & hip.FLAG_F16X2 else 0.0)
                same_inf = torch.equal(torch.isinf(out), torch.isinf(ref))
                d = (out - ref)[torch.isfinite(ref)]
                note(name, (d.abs().max().item() / t if d.numel() else 0.0) if same_inf else float("inf"))

            # gradient in x of sum_i softmin_i, float32 clouds
            if x.dtype == torch.float32 and eps >= 1e-2 * cost:
                xb, yb, hb = (t if B is not None else t[None] for t in (x.contiguous(), y.contiguous(), h.contiguous()))
                truth = hip.softmin(eps, xb, yb, hb, p=p, flags=hip.FLAG_DIRECT)
                ones = torch.ones_like(truth)
                gref = hip.softmin_bwd_x_raw(xb, yb, hb, truth, ones, eps, p=p, flags=hip.FLAG_DIRECT)
                for name, flags in [(f"p{p} softmin gradient", 0)] + ([("p2 softmin gradient f16x2", hip.FLAG_F16X2)] if p == 2 and diam2 / eps < 1e5 else []):
                    got = hip.softmin_bwd_x_raw(xb, yb, hb, truth, ones, eps, p=p, flags=flags)
                    # an exponent error delta = 4e-7 cost / eps (the forward tolerance) moves every plan weight by delta, and the weights
                    # multiply dC/dx: x - y (p = 2, up to a diameter) or a unit vector (p = 1) — not the gradient they sum to
                    gt = 5e-5 * gref.abs().max().item() + 3 * 4e-7 * cost / eps * (diam if p == 2 else 1.0) + 1e-30
                    if p == 1:
                        gt += 2e-3      # unit vectors of near pairs: directions of differences of ~1e-3 diam, see DESIGN.md (distance kernels)
                    note(name, (got - gref).abs().max().item() / gt)

        # ---- laplacian / energy products and gradients ----
        v = torch.tensor(rng.standard_normal(shp(M)[:-1]) / M, dtype=torch.float32, device=dev)
        blur = float(np.sqrt(eps))
        for kind in ("laplacian", "energy", "gaussian"):
            refc = hip.kernel_conv(kind, x, y, v, blur, flags=hip.FLAG_NO_MFMA)
            bound = hip.kernel_conv(kind, x, y, v.abs(), blur, flags=hip.FLAG_NO_MFMA).abs().max().item()
            expansion = 2.4e-7 * diam2 / blur**2 if kind == "gaussian" else 2e-6 * diam / blur if kind == "laplacian" else 0.0
            out = hip.kernel_conv(kind, x, y, v, blur)
            floor = 2e-6 * diam * v.abs().sum(-1).max().item() if kind == "energy" else 0.0
            note(f"{kind} product", (out - refc).abs().max().item() / ((3e-6 + expansion) * bound + floor + 1e-30), blur=blur)
            if x.dtype == torch.float32:
                res = {}
                for fl in (0, hip.FLAG_NO_MFMA):
                    xg = x.clone().requires_grad_(True)
                    (gx,) = torch.autograd.grad(hip.kernel_conv(kind, xg, y, v.abs(), blur, flags=fl).sum(), [xg])
                    res[fl] = gx
                gmax = res[hip.FLAG_NO_MFMA].abs().max().item()
                # energy: d|x-y| is a unit vector per pair, near pairs carry a direction error of ~1e-3 each
                # gaussian: the gradient is x_i S0 - S1, a difference of terms of size |x| sum |v| k / blur^2 (zero gradients of coincident
                # or far-away points come out as the rounding of that difference)
                vsum = v.abs().sum(-1).max().item()
                slack = 2e-3 * vsum if kind != "gaussian" else 4e-7 * diam / blur**2 * vsum
                note(f"{kind} gradient", (res[0] - res[hip.FLAG_NO_MFMA]).abs().max().item() / ((3e-5 + 4 * expansion) * gmax + slack + 1e-30), blur=blur)

        # ---- the fused iteration against four single steps ----
        if True:
            xb, yb = (t if B is not None else t[None] for t in (x.contiguous(), y.contiguous()))
            Bn = xb.shape[0]
            a_log = torch.full((Bn, N), -float(np.log(N)), device=dev)
            b_log = torch.full((Bn, M), -float(np.log(M)), device=dev)
            for p in (2, 1):
                cost = diam2 if p == 2 else diam
                if not (1e-3 * cost <= eps):
                    continue
                pots0 = [torch.tensor(rng.standard_normal((Bn, n)) * 0.1 * cost, dtype=torch.float32, device=dev) for n in (N, M, N, M)]
                for fl in ([0, hip.FLAG_F16X2] if p == 2 and diam2 / eps < 1e5 else [0]):
                    f_ba, g_ab, f_aa, g_bb = pots0
                    ex = [hip.sinkhorn_step(eps, xb, yb, b_log, g_ab, f_ba, 0.5, p=p, flags=fl),
                          hip.sinkhorn_step(eps, yb, xb, a_log, f_ba, g_ab, 0.5, p=p, flags=fl),
                          hip.sinkhorn_step(eps, xb, xb, a_log, f_aa, f_aa, 0.5, p=p, flags=fl),
                          hip.sinkhorn_step(eps, yb, yb, b_log, g_bb, g_bb, 0.5, p=p, flags=fl)]
                    got = hip.sinkhorn_iter4(eps, xb, yb, a_log, b_log, [t.clone() for t in pots0], 0.5, debias=True, flags=fl, p=p)
                    # the exponent offsets h = log w + pot / eps are formed in float32 inside both kernels (orders differ): eps ulp(h)
                    hmax = float(np.log(max(N, M))) + max(t.abs().max().item() for t in pots0) / eps
                    t = (4e-7 * cost + 2e-6 * max(e.abs().max().item() for e in ex) + (1e-3 * diam * 2e-3 if p == 1 else 0.0)
                         + 2.4e-7 * hmax * eps)
                    note(f"p{p} iter4" + (" f16x2" if fl else ""), max((a - b).abs().max().item() for a, b in zip(got, ex)) / t)

    bad = False
    for k, (e, c) in sorted(worst.items()):
        print(f"{k:28s} worst error / tolerance = {e:.3f}   at {c}")
        bad |= not (e <= 1.0)
    print("cases:", n_cases, "FAIL" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
