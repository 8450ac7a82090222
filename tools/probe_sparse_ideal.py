"""GPU box: the block-sparse soft-min forward on SYNTHETIC patterns — clusters of exactly 512 (or 455) sorted rows, every row cluster
keeping 461 column clusters as (a) one run, (b) 66 runs of 7 — against the real two-scale pattern's 20 ms per 2.1e11 pairs: what
irregular clusters, partial tiles and short runs cost.  usage: probe_sparse_ideal.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip
from geomloss_amd.cluster import from_matrix
dev = torch.device("cuda:0")

def run(csize, runs, label, flags=hip.FLAG_F16X2):
    C = 2048
    N = C * csize
    g = torch.Generator().manual_seed(0)
    x, y = torch.rand(N, 3, generator=g).to(dev), torch.rand(N, 3, generator=g).to(dev)
    h = torch.zeros(N, device=dev)
    r = torch.arange(C, device=dev, dtype=torch.int32)
    ranges = torch.stack([r * csize, (r + 1) * csize], 1).contiguous()
    i = torch.arange(C, device=dev)[:, None]
    j = torch.arange(C, device=dev)[None, :]
    d = (j - i) % C
    if runs == 1:
        keep = d < 461
    else:                                   # `runs` runs of 7 clusters, 31 apart
        keep = ((d % 31) < 7) & (d < 31 * runs)
    kept = int(keep.sum().item()) * csize * csize
    rg = from_matrix(ranges, ranges, keep)
    eps = 1.0
    for _ in range(2):
        hip.softmin(eps, x, y, h, ranges=rg, flags=flags)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        hip.softmin(eps, x, y, h, ranges=rg, flags=flags)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{label}: cluster {csize}, kept pairs {kept:.3e}: {ms:.2f} ms = {kept / ms / 1e9:.2f}e12 pairs/s  ({kept / ms / 1e9 / 15.73:.3f} of 1.573e13)", flush=True)

run(512, 1, "one run of 461 clusters  ")
run(512, 66, "66 runs of 7 clusters     ")
run(455, 66, "66 runs of 7, 455 rows    ")
run(455, 1, "one run, 455 rows         ")
run(512, 66, "66 runs of 7, bf16 x 3    ", flags=0)
for cs in (530, 480, 448, 416, 384):      # 4 x 128 + 18 rows; 3 x 128 + 96 / + 64 / + 32 / + 0 rows
    run(cs, 1, f"one run, {cs} rows         ")
