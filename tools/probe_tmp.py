import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for n in (20000, 1000000):
    x = torch.rand(n, 3, generator=g).to(dev); y = torch.rand(n, 3, generator=g).to(dev)
    for grad in (False, True):
        xx = x.clone().requires_grad_(grad)
        for kw in (dict(backend="online"), dict(backend="multiscale", truncate=3), dict(backend="multiscale", truncate=1)):
            L = SamplesLoss("gaussian", blur=0.1, **kw)(xx, y)
            print(n, grad, kw, repr(L.item()))
