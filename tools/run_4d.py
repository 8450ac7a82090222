"""The reference's 4-D recipe (examples/sinkhorn_multiscale/plot_optimal_transport_cluster.py:155-166): clusters given as labels — voxels
of the 3 spatial coordinates of (position, feature) points — and the two-scale solver on the 4-D clouds; for kernel traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000


def labels4d(t, scale=0.08):
    q = (t[:, :3] / scale).floor().long()
    return torch.unique((q[:, 0] * 64 + q[:, 1]) * 64 + q[:, 2], return_inverse=True)[1].int()


g = torch.Generator().manual_seed(3)
x4, y4 = torch.rand(n, 4, generator=g).to(dev), torch.rand(n, 4, generator=g).to(dev)
loss4 = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="multiscale")
w4 = torch.full((n,), 1.0 / n, device=dev)
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L = loss4(labels4d(x4), w4, x4, labels4d(y4), w4, y4)
    torch.cuda.synchronize()
    print(f"rep {r}: {time.perf_counter() - t0:.4f} s  loss {L.item():.6e}", flush=True)
