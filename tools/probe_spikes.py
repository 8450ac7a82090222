import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geomloss_amd import SamplesLoss, hip
dev = torch.device("cuda:0")
x, y = bench.cfg4_batch(dev, 32, seed=2)
h = torch.zeros(32, 4096, device=dev)
def timeit(name, fn, n=16):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:46s}", " ".join(f"{t:.1f}" for t in ts), flush=True)
def raw40():
    for _ in range(40): hip.softmin_fwd_raw(x, y, h, 0.01, 2)
timeit("40 raw softmin_fwd (ws alloc each)", raw40)
lib = hip.load_library()
nb = int(lib.glhip_workspace_bytes(32, 4096, 4096, 3, 0)); ws = torch.empty(nb, dtype=torch.uint8, device=dev); out = torch.empty(32, 4096, device=dev)
print("workspace bytes", nb)
st = torch.cuda.current_stream().cuda_stream
def raw40_fixed():
    for _ in range(40):
        lib.glhip_softmin_fwd(x.data_ptr(), y.data_ptr(), h.data_ptr(), out.data_ptr(), 32, 4096, 4096, 3, 0.01, 2, hip.BF16, None, None, None, 0, ws.data_ptr(), nb, 0, st)
timeit("40 raw ctypes calls, fixed workspace", raw40_fixed)
def raw40_nows():
    for _ in range(40):
        lib.glhip_softmin_fwd(x.data_ptr(), y.data_ptr(), h.data_ptr(), out.data_ptr(), 32, 4096, 4096, 3, 0.01, 2, hip.BF16, None, None, None, 0, None, 0, 0, st)
timeit("40 raw ctypes calls, no workspace", raw40_nows)
L = SamplesLoss("sinkhorn", backend="online", **bench.CFG4)
timeit("full loss", lambda: L(x, y))
gc.disable()
timeit("full loss, gc disabled", lambda: L(x, y))
gc.enable()
os.environ["X"] = "1"
def torch_only():
    a = torch.empty(nb, dtype=torch.uint8, device=dev)
    for _ in range(150): a[:1024].add_(1)
timeit("150 tiny torch kernels", torch_only)
