"""Where the time of the matrix-core distance kernels goes at N = M = 1e6: voxel row blocks (what hip.kernel_conv uses) against
fixed 256-row slabs of the same voxel-sorted cloud (every workgroup full), the number of column chunks, the voxel size, and the
near-pair guard switched off (GLHIP_DIST_GUARD=0 in the environment).   usage: python tools/probe_dist_blocks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import hip
from bench import event_ms, make_problem

n = 1_000_000
dev = torch.device("cuda:0")
x, y, h, eps = make_problem(n, dev, seed=7)
v = torch.rand(1, n, device=dev) / n


def time_plan(plan, label):
    vs = plan.cols(v)
    hs = plan.cols(h)
    t_e = event_ms(lambda: hip.kernel_conv_fwd_raw(hip.ENERGY, plan.x, plan.y, vs, 0.05, plan.ranges, hip.FLAG_MFMA_DIST), 3)
    t_l = event_ms(lambda: hip.kernel_conv_fwd_raw(hip.LAPLACIAN, plan.x, plan.y, vs, 0.05, plan.ranges, hip.FLAG_MFMA_DIST), 2)
    t_s = event_ms(lambda: hip.softmin_fwd_raw(plan.x, plan.y, hs, 0.05, 1, plan.ranges, hip.FLAG_MFMA_DIST), 2)
    sizes = (plan.ranges.ranges_i[:, 1] - plan.ranges.ranges_i[:, 0]).float()
    print(f"{label:58s}: energy {t_e:7.2f}  laplacian {t_l:7.2f}  soft-min p=1 {t_s:7.2f} ms   ({sizes.numel()} row blocks, "
          f"{sizes.mean():.0f} +- {sizes.std():.0f} rows, max {int(sizes.max())})", flush=True)


t_sort = event_ms(lambda: hip._CompactRows(x, y), 3)
print(f"building the plan (two voxel sorts + gathers): {t_sort:.2f} ms")
for rows in (256, 224, 200, 128):
    hip._DIST_ROWS_PER_VOXEL = rows
    time_plan(hip._CompactRows(x, y), f"voxel row blocks, ~{rows} rows per voxel, 8 column chunks")
hip._DIST_ROWS_PER_VOXEL = 256
for chunks in (3, 6, 24):
    hip._DIST_COL_CHUNKS = chunks
    time_plan(hip._CompactRows(x, y), f"voxel row blocks, ~256 rows per voxel, {chunks} column chunks")
hip._DIST_COL_CHUNKS = 8
plan = hip._CompactRows(x, y)
C = (n + 255) // 256
starts = torch.arange(C, device=dev, dtype=torch.int32) * 256
slabs = torch.stack((starts, (starts + 256).clamp_max(n)), 1).contiguous()
nchunk = plan.ranges.redranges_j.shape[0] // plan.ranges.ranges_i.shape[0]
red = plan.ranges.redranges_j[:nchunk].repeat(C, 1).contiguous()
slices = (torch.arange(1, C + 1, device=dev, dtype=torch.int32) * nchunk).contiguous()
plan.ranges = hip.BlockRanges(slabs, slices, red, None, None, None)
time_plan(plan, "fixed 256-row slabs of the voxel-sorted cloud (full workgroups)")
