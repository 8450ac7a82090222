"""BASELINE configs[2] at its stated size: SamplesLoss("sinkhorn", backend="multiscale") at N = M = 1e6, 3D fp32, end to end
against the float64 two-scale oracle (oracle/oracle_torch64.py, fine level cluster by cluster on the GPU).  Test infrastructure
(imports oracle/): run on the GPU box, paste the output into profiles/.   usage: python tools/verify_cfg3.py [N] [same|shift]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from geomloss_amd import SamplesLoss
from oracle import oracle_torch64 as o64

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "same"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x, y = torch.rand(N, 3, generator=g).to(dev), torch.rand(N, 3, generator=g).to(dev)
if kind == "shift":
    y = y * 0.6 + 0.3
kw = dict(p=2, blur=0.05)
xg = x.clone().requires_grad_(True)
L = SamplesLoss("sinkhorn", backend="multiscale", **kw)(xg, y)
(gx,) = torch.autograd.grad(L, [xg])
F, G = SamplesLoss("sinkhorn", backend="multiscale", potentials=True, **kw)(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
a = np.full(N, 1.0 / N)
(ref, ref_gx), info = o64.sinkhorn_multiscale(a, x, a, y, grad=True, return_info=True, device=dev, **kw)
Fo, Go = o64.sinkhorn_multiscale(a, x, a, y, potentials=True, device=dev, **kw)
t1 = time.perf_counter()
rel = lambda u, v: float(np.abs(u - v).max() / np.abs(v).max())
print(f"cfg3 N=M={N} ({kind}-law clouds): HIP loss {L.item():.9e}  float64 two-scale oracle {ref:.9e}  rel {abs(L.item() - ref) / abs(ref):.2e}")
print(f"  dL/dx max-norm rel {rel(gx.cpu().numpy(), ref_gx):.2e};  potentials F abs {np.abs(F.cpu().numpy() - Fo).max():.2e} (scale {np.abs(Fo).max():.2e}), "
      f"G abs {np.abs(G.cpu().numpy() - Go).max():.2e}")
print(f"  clusters {info['n_clusters']}, jump after iteration {info['jumps'][0]} of {len(info['eps_list'])}, kept fractions {[round(k, 4) for k in info['kept_fraction']]}; "
      f"oracle time {t1 - t0:.0f} s")
