"""Randomised cross-check of the matrix-core kernels against the direct-difference VALU kernels (GPU).

Shapes straddle the tile (512 columns), group (32) and row-block (128 / 256) boundaries; eps spans 5 decades; dual
vectors include large offsets and -1e5 entries.  Prints the worst normalised error per code path; exits non-zero on a
violation of the tolerance used by tests/test_hip_kernels.py.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from geomloss_amd import hip


def main(n_cases=300, seed=0):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    worst = {}
    edges = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 1536, 2047, 2049]
    for case in range(n_cases):
        pick = lambda: int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 6000))  # noqa: E731
        N, M, D = pick(), pick(), int(rng.integers(1, 4))
        B = None if rng.random() < 0.7 else int(rng.integers(2, 5))
        eps = float(10 ** rng.uniform(-4, 1))
        shp = (lambda n: (n, D)) if B is None else (lambda n: (B, n, D))
        scale = float(10 ** rng.uniform(-1, 1))
        x = torch.tensor(rng.random(shp(N)) * scale, dtype=torch.float32, device=dev)
        y = torch.tensor(rng.random(shp(M)) * scale, dtype=torch.float32, device=dev)
        h = rng.standard_normal(shp(M)[:-1]) * rng.choice([0.1, 3.0, 50.0])
        if rng.random() < 0.3:
            h[..., rng.integers(0, M)] = -100000.0
        if rng.random() < 0.2:
            h = h + np.arange(M) * rng.choice([0.5, -0.5])      # drifting maximum: exercises the lazy-max redo
        h = torch.tensor(h, dtype=torch.float32, device=dev)
        if rng.random() < 0.25:
            x, y = x.bfloat16(), y.bfloat16()
        ref = hip.softmin(eps, x, y, h, flags=hip.FLAG_DIRECT)
        diam2 = D * scale * scale
        tol = 4e-7 * diam2 + 2e-6 * ref.abs().max().item() + 1e-30
        variants = [("x32", 0), ("x32+prepack", hip.FLAG_PREPACK), ("x32 nosplit", hip.FLAG_NO_SPLIT),
                    ("xdl16", hip.FLAG_XDL16), ("f32 mfma", hip.FLAG_F32_MFMA)]
        if diam2 / eps < 1e5:      # the f16 x 2 exponent layout inside the range its flag vouches for (include/glhip.h)
            variants += [("f16x2", hip.FLAG_F16X2), ("f16x2+prepack", hip.FLAG_F16X2 | hip.FLAG_PREPACK), ("f16x2 nosplit", hip.FLAG_F16X2 | hip.FLAG_NO_SPLIT)]
        for name, flags in variants:
            out = hip.softmin(eps, x, y, h, flags=flags)
            # f16 x 2: the scalar item of an exponent (H_j, the running maximum) is three f16 pieces of H / 8 — an ABSOLUTE floor of
            # 8 x 2^-25 = 2.4e-7 per exponent, i.e. 2.4e-7 eps ln 2 on a potential (csrc/glhip_klayout.h), which only shows where eps is
            # large and the values are small (the first temperatures of an annealing loop)
            t = tol + (2 * 2.4e-7 * eps * 0.6931 if flags & hip.FLAG_F16X2 else 0.0)
            err = (out - ref).abs().max().item() / t
            if not np.isfinite(err) and torch.equal(torch.isinf(out), torch.isinf(ref)):
                err = 0.0
            if err > worst.get(name, (0,))[0]:
                worst[name] = (err, dict(N=N, M=M, D=D, B=B, eps=eps, scale=scale, dtype=str(x.dtype)))
        # gaussian product, same shapes
        v = torch.tensor(rng.standard_normal(shp(M)[:-1]) / M, dtype=torch.float32, device=dev)
        blur = float(np.sqrt(eps))
        refc = hip.kernel_conv("gaussian", x, y, v, blur, flags=hip.FLAG_NO_MFMA)
        bound = hip.kernel_conv("gaussian", x, y, v.abs(), blur, flags=hip.FLAG_NO_MFMA)
        tolc = (3e-6 + 2.4e-7 * diam2 / blur**2) * bound.abs().max().item() + 1e-30   # signed weights: errors scale with sum |v| k
        for name, flags in (("gauss x32", 0), ("gauss x32+prepack", hip.FLAG_PREPACK), ("gauss xdl16", hip.FLAG_XDL16)):
            out = hip.kernel_conv("gaussian", x, y, v, blur, flags=flags)
            err = (out - refc).abs().max().item() / tolc
            if err > worst.get(name, (0,))[0]:
                worst[name] = (err, dict(N=N, M=M, D=D, B=B, blur=blur, scale=scale, dtype=str(x.dtype)))
        # one-pass kernels (round 2) against the two reductions they replace, same shapes
        def note(name, err, **cfg):
            if not (err <= worst.get(name, (0,))[0]):
                worst[name] = (err, dict(N=N, M=M, D=D, B=B, scale=scale, dtype=str(x.dtype), **cfg))

        vp = v.abs() + 1e-3 / M
        for kind in ("gaussian", "laplacian", "energy"):
            res = {}
            for fused in (True, False):
                hip.set_kernel_grad_fusion(fused)
                xg = x.clone().requires_grad_(True)
                out = hip.kernel_conv(kind, xg, y, vp, blur)
                (gx,) = torch.autograd.grad(out.sum(), [xg])
                res[fused] = (out.detach(), gx.float())
            hip.set_kernel_grad_fusion(True)
            expansion = 2.4e-7 * diam2 / blur**2 if kind == "gaussian" else 0.0
            note(f"{kind} one-pass product", (res[True][0] - res[False][0]).abs().max().item()
                 / ((6e-6 + 2 * expansion) * res[False][0].abs().max().item() + 1e-30), blur=blur)
            if x.dtype == torch.float32:     # bf16 clouds: the gradient is rounded to bf16 on the way out
                note(f"{kind} one-pass gradient", (res[True][1] - res[False][1]).abs().max().item()
                     / ((3e-5 + 4 * expansion) * res[False][1].abs().max().item() + 1e-30), blur=blur)
        if eps >= 1e-3:
            xb, yb, hb = (t if B is not None else t[None] for t in (x.contiguous(), y.contiguous(), h))
            hb = hb.contiguous()
            truth = hip.softmin(eps, xb, yb, hb)
            ones = torch.ones_like(truth)
            gref = hip.softmin_bwd_x_raw(xb, yb, hb, truth, ones, eps)
            margin = float(rng.choice([1e-3, 1.0, 20.0])) * eps
            guess = truth + torch.tensor(rng.uniform(-margin, margin, tuple(truth.shape)), dtype=torch.float32, device=dev)
            out, unit = hip.softmin_fwd_grad_raw(xb, yb, hb, guess.contiguous(), margin * 1.01 + 1e-6 * truth.abs().max().item(), eps)
            # value = guess + (a correction of up to 2 margins): a few ulps of the margin on top of the forward tolerance
            note("softmin value+grad: value", (out.reshape(ref.shape) - ref).abs().max().item() / (tol + 4e-7 * margin), eps=eps, margin=margin)
            if x.dtype == torch.float32:
                note("softmin value+grad: gradient", (unit - gref).abs().max().item()
                     / ((5e-5 + 2 * 4e-7 * diam2 / eps) * gref.abs().max().item() + 1e-30), eps=eps, margin=margin)
                # (an exponent error delta = 4e-7 diam^2 / eps — the forward tolerance above — moves the weights by delta)
    bad = False
    for k, (e, cfg) in sorted(worst.items()):
        print(f"{k:20s} worst error / tolerance = {e:.3f}   at {cfg}")
        bad |= not (e <= 1.0)
    print("cases:", n_cases, "FAIL" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
