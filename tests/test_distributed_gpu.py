"""Two ranks on ONE GPU: the batch-sharded loss with device tensors and the HIP `online` backend in each process, collectives over
`gloo` (which stages CUDA tensors through the host).  RCCL refuses two ranks on one device, and no multi-GPU node was available in any
round — this is the closest the single-GPU boxes get to the N > 1 leg of bench.py: two processes with their own HIP library and
streams, the global bounding-box exchange (MIN / MAX all-reduce of device tensors), the scalar SUM all-reduce, uneven shards, an empty
rank, gradients of the local shard.  The same module runs on RCCL when `init_process_group("nccl")` is given one GPU per rank."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geomloss_amd import SamplesLoss, hip
        from geomloss_amd.distributed import ShardedSamplesLoss, shard_batch, shard_bounds

        dev = torch.device("cuda:0")
        hip.load_library()                       # the product path: fails loudly without the HIP extension
        g = torch.Generator().manual_seed(0)
        N, M = 700, 900
        x = torch.rand(B, N, 3, generator=g).to(dev)
        y = (torch.rand(B, M, 3, generator=g) * torch.linspace(0.6, 1.4, B)[:, None, None]).to(dev)      # heterogeneous boxes
        base = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online")
        xf = x.clone().requires_grad_(True)
        full = base(xf, y)                       # (B,) unsharded: one schedule from the global bounding box
        (gf,) = torch.autograd.grad(full.sum(), [xf])

        lo, hi = shard_bounds(B, rank, world)
        xl = shard_batch(x).clone().requires_grad_(True)
        yl = shard_batch(y)
        total = ShardedSamplesLoss(base, "sum")(xl, yl)
        gerr = 0.0
        if hi > lo:
            (gl,) = torch.autograd.grad(total, [xl])
            gerr = float((gl - gf[lo:hi]).abs().max() / gf.abs().max())
        vec = ShardedSamplesLoss(base, "none")(xl.detach(), yl)
        q.put((rank, float((total - full.sum()).abs() / full.sum().abs()), gerr, float((vec - full).abs().max() / full.abs().max()),
               str(total.device), None))
    except Exception as e:      # surface the failure in the parent
        q.put((rank, None, None, None, None, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 1])      # 5: ranks own 3 and 2 problems; 1: rank 1 owns none and still takes part in every collective
def test_sharded_loss_on_device_tensors_two_ranks_one_gpu(cuda, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, rel_total, gerr, verr, device, err in results:
        assert err is None, f"rank {rank}: {err}"
        assert device.startswith("cuda")
        # the two processes run the same kernels on the same inputs: the sharded and the unsharded loss agree to float32 rounding
        assert rel_total < 1e-5 and gerr < 1e-4 and verr < 1e-5, (rank, rel_total, gerr, verr)
