"""Quick kernel timings on one GPU (development aid; bench.py is the contract)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip


def timeit(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return min(ts), sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="100000,1000000")
    ap.add_argument("--what", default="softmin,softmin_direct,softmin_p1,bwd,gauss,lap,energy")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs")
    if "batched" in args.what:
        for (B, N, dt) in ((256, 4096, torch.bfloat16), (256, 4096, torch.float32), (64, 16384, torch.float32), (1, 50000, torch.float32)):
            torch.manual_seed(0)
            x, y = torch.rand(B, N, 3, device=dev).to(dt), torch.rand(B, N, 3, device=dev).to(dt)
            h = torch.full((B, N), -float(torch.log(torch.tensor(float(N)))), device=dev)
            for fl, nm in ((0, "32x32x16"), (16, "16x16x32")):
                t = timeit(lambda: hip.softmin_fwd_raw(x, y, h, 0.05**2, 2, flags=fl), reps=5)
                print("B=%d N=M=%d %s softmin fwd %s: %.3f ms  %.3e pairs/s" % (B, N, str(dt)[6:], nm, t[0] * 1e3, B * N * N / t[0]))
        return
    for N in [int(s) for s in args.sizes.split(",")]:
        torch.manual_seed(0)
        x, y = torch.rand(N, 3, device=dev), torch.rand(N, 3, device=dev)
        eps = 0.05**2
        h = torch.full((N,), -float(torch.log(torch.tensor(float(N)))), device=dev) + 0.01 * torch.randn(N, device=dev) / eps
        xb, yb, hb = x[None], y[None], h[None]
        pairs = float(N) * N
        what = args.what.split(",")
        res = {}
        if "fwd" in what:
            res["softmin p2 (bf16x3 32x32x16)"] = timeit(lambda: hip.softmin_fwd_raw(xb, yb, hb, eps, 2), reps=5)
            res["softmin p2 (bf16x3 16x16x32)"] = timeit(lambda: hip.softmin_fwd_raw(xb, yb, hb, eps, 2, flags=16), reps=5)
        if "softmin" in what:
            res["softmin p2 (xdl bf16x3)"] = timeit(lambda: hip.softmin_fwd_raw(xb, yb, hb, eps, 2))
            res["softmin p2 f32 mfma"] = timeit(lambda: hip.softmin_fwd_raw(xb, yb, hb, eps, 2, flags=8))
            res["softmin p2 mfma nosplit"] = timeit(lambda: hip.softmin_fwd_raw(xb, yb, hb, eps, 2, flags=4))
            res["softmin p2 valu split"] = timeit(lambda: hip.softmin_fwd_raw(xb, yb, hb, eps, 2, flags=2))
            res["softmin p2 valu nosplit"] = timeit(lambda: hip.softmin_fwd_raw(xb, yb, hb, eps, 2, flags=6))
        if "softmin_direct" in what:
            res["softmin p2 direct"] = timeit(lambda: hip.softmin_fwd_raw(xb, yb, hb, eps, 2, flags=1))
        if "softmin_p1" in what:
            res["softmin p1"] = timeit(lambda: hip.softmin_fwd_raw(xb, yb, hb, 0.05, 1))
        if "bwd" in what:
            out = hip.softmin_fwd_raw(xb, yb, hb, eps, 2)
            g = torch.ones_like(out)
            res["softmin bwd p2 (mfma)"] = timeit(lambda: hip.softmin_bwd_x_raw(xb, yb, hb, out, g, eps, 2))
            res["softmin bwd p2 valu"] = timeit(lambda: hip.softmin_bwd_x_raw(xb, yb, hb, out, g, eps, 2, flags=2))
        v = torch.full((1, N), 1.0 / N, device=dev)
        for nm, kind in (("gauss", 0), ("lap", 1), ("energy", 2)):
            if nm in what:
                res["conv " + nm] = timeit(lambda: hip.kernel_conv_fwd_raw(kind, xb, yb, v, 0.05))
                if nm == "gauss":
                    res["conv gauss 16x16x32"] = timeit(lambda: hip.kernel_conv_fwd_raw(kind, xb, yb, v, 0.05, flags=16))
                    res["conv gauss (again)"] = timeit(lambda: hip.kernel_conv_fwd_raw(kind, xb, yb, v, 0.05))
                    res["conv gauss valu"] = timeit(lambda: hip.kernel_conv_fwd_raw(kind, xb, yb, v, 0.05, flags=2))
                    gg = torch.ones(1, N, device=dev)
                    res["conv gauss bwd (mfma)"] = timeit(lambda: hip.kernel_conv_bwd_x_raw(kind, xb, yb, v, gg, 0.05))
                    res["conv gauss bwd valu"] = timeit(lambda: hip.kernel_conv_bwd_x_raw(kind, xb, yb, v, gg, 0.05, flags=2))
        if "dense" in what and N <= 30000:
            C = torch.rand(1, N, N, device=dev)
            tmin, tmed = timeit(lambda: hip.softmin_dense_fwd_raw(C, hb, eps))
            print(f"N=M={N:>8d} {'dense-matrix softmin':26s} min {tmin*1e3:10.3f} ms  {N*N*4/tmin/1e9:8.1f} GB/s of matrix streamed")
        for k, (tmin, tmed) in res.items():
            print(f"N=M={N:>8d} {k:26s} min {tmin*1e3:10.3f} ms  med {tmed*1e3:10.3f} ms  {pairs/tmin:.3e} pairs/s")


if __name__ == "__main__":
    main()
