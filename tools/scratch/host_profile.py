"""GPU box: cProfile of the host side of a small online / multiscale loss (where the interpreter's time goes between launches).
usage: host_profile.py [N] [backend]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geomloss_amd import SamplesLoss
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
backend = sys.argv[2] if len(sys.argv) > 2 else "online"
g = torch.Generator().manual_seed(3)
x, y = torch.rand(n, 3, generator=g).to(dev), torch.rand(n, 3, generator=g).to(dev)
loss = SamplesLoss("sinkhorn", p=2, blur=0.05, backend=backend)
for _ in range(20):
    loss(x, y)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    loss(x, y)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.print_callers("view")
