"""Value-only kernel norms at N = M = 1e6: upper-triangle self-terms against the three full products and the float64 oracle
(oracle/oracle_hip64), two samples of one law (loss ~1e-6 of terms ~0.5: the hard case) and shifted clouds.  Seconds + values."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geomloss_amd.kernel_samples as ks
from geomloss_amd import SamplesLoss
from oracle import oracle_torch64 as o64
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
with_oracle = os.environ.get("ORACLE", "1") != "0"
g = torch.Generator().manual_seed(1)
x, y0 = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
a = torch.full((n,), 1.0 / n, device=dev)


def timed(f):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v = f()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts[1:]), v.item()


for case, y in (("same law", y0), ("shifted", y0 * 0.8 + 0.15)):
    for name, kw in (("energy", {}), ("laplacian", dict(blur=0.05)), ("gaussian", dict(blur=0.05))):
        loss = SamplesLoss(name, backend="online", **kw)
        ks._UPPER_MIN_PAIRS = 2e9
        t_half, L_half = timed(lambda: loss(x, y))
        ks._UPPER_MIN_PAIRS = float("inf")
        t_full, L_full = timed(lambda: loss(x, y))
        ref = o64.kernel_loss(name, x, y, a, a, blur=kw.get("blur", 0.05), device=dev) if with_oracle else float("nan")
        print(f"{case:9s} {name:10s} upper {t_half:.4f} s  full {t_full:.4f} s   loss upper {L_half:.6e}  full {L_full:.6e}  fp64 {ref:.6e}", flush=True)
