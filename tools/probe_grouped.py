"""GPU box: would ROW GROUPS pay?  A block-sparse soft-min whose row blocks are windows of W sorted rows (several small clusters) reducing
over the UNION of their clusters' kept column intervals, against one row block per cluster (the reference's pattern) and the dense
launch.  Keep rule: |c_i - c_j| <= r with r such that ~21 % of the matrix is kept (what the dual-slack rule keeps in 3-D, any N)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip, sinkhorn_samples as ss
from geomloss_amd.cluster import from_matrix
dev = torch.device("cuda:0")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


for n in (10_000, 30_000, 100_000, 200_000):
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
    w = torch.full((n,), 1.0 / n, device=dev)
    scale = 3 ** 0.5 / (3 ** 0.5 * 2000 ** (1 / 3))
    [a_c, a_s], [x_c, x_s], [rx], _ = ss.clusterize(w, x, scale=scale)
    [b_c, b_s], [y_c, y_s], [ry], _ = ss.clusterize(w, y, scale=scale)
    keep = torch.cdist(x_c, y_c) <= 0.36
    h = torch.zeros(n, device=dev)
    eps = 0.05 ** 2
    rows = (rx[:, 1] - rx[:, 0]).double()
    cols = (ry[:, 1] - ry[:, 0]).double()
    kept = float(rows @ (keep.double() @ cols)) / n ** 2
    rg = from_matrix(rx, ry, keep)
    t_c, out_c = timed(lambda: hip.softmin(eps, x_s, y_s, h, ranges=rg))
    t_d, out_d = timed(lambda: hip.softmin(eps, x_s, y_s, h))
    line = f"N = {n}: clusters {x_c.shape[0]} (mean {n / x_c.shape[0]:.0f} rows), kept {kept:.3f}: per cluster {t_c * 1e6:8.1f} us   dense {t_d * 1e6:8.1f} us"
    for W in (64, 128, 256):
        gid = (rx[:, 0] // W).long()
        G = (n + W - 1) // W
        kg = torch.zeros(G, keep.shape[1], device=dev, dtype=torch.int32).index_add_(0, gid, keep.int()) > 0
        start = torch.full((G,), n, device=dev, dtype=torch.int32).scatter_reduce_(0, gid, rx[:, 0].int(), "amin")
        end = torch.zeros(G, device=dev, dtype=torch.int32).scatter_reduce_(0, gid, rx[:, 1].int(), "amax")
        live = end > start
        rgx = torch.stack([start, end], 1)[live].contiguous()
        kg = kg[live]
        keptg = float((rgx[:, 1] - rgx[:, 0]).double() @ (kg.double() @ cols)) / n ** 2
        rgg = from_matrix(rgx, ry, kg)
        t_g, out_g = timed(lambda: hip.softmin(eps, x_s, y_s, h, ranges=rgg))
        line += f"   W={W}: {t_g * 1e6:8.1f} us (kept {keptg:.3f}, |d| {float((out_g - out_c).abs().max()):.1e})"
    print(line, flush=True)
