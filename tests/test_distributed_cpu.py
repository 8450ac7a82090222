"""World-size-2 `gloo` test of the batch-sharded loss (the N>1 path of bench.py uses the same module on RCCL)."""

import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geomloss_amd import SamplesLoss
        from geomloss_amd.distributed import ShardedSamplesLoss, shard_batch, shard_bounds

        torch.manual_seed(0)
        B, N, M = 5, 40, 50   # odd batch: ranks own 3 and 2 items
        x = torch.rand(B, N, 2)
        y = torch.rand(B, M, 2) * torch.linspace(0.5, 1.5, B)[:, None, None]   # heterogeneous boxes
        base = SamplesLoss("sinkhorn", p=2, blur=0.1, backend="tensorized")
        full = base(x, y)   # unsharded reference: one schedule from the global bounding box

        xl = shard_batch(x).clone().requires_grad_(True)
        yl = shard_batch(y)
        lo, hi = shard_bounds(B, rank, world)
        total = ShardedSamplesLoss(base, "sum")(xl, yl)
        (g,) = torch.autograd.grad(total, [xl])
        xf = x.clone().requires_grad_(True)
        (gf,) = torch.autograd.grad(base(xf, y).sum(), [xf])
        vec = ShardedSamplesLoss(base, "none")(xl.detach(), yl)
        mean = ShardedSamplesLoss(base, "mean")(xl.detach(), yl)
        q.put((rank, float((total - full.sum()).abs() / full.sum().abs()), float((g - gf[lo:hi]).abs().max()),
               float((vec - full).abs().max()), float((mean - full.mean()).abs()), None))
    except Exception as e:  # surface the failure in the parent
        q.put((rank, None, None, None, None, repr(e)))
    finally:
        dist.destroy_process_group()


def test_sharded_loss_equals_unsharded_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, rel_total, gerr, verr, merr, err in results:
        assert err is None, f"rank {rank}: {err}"
        assert rel_total < 1e-6 and gerr < 1e-6 and verr < 1e-6 and merr < 1e-6


def _worker_short_batch(rank, world, port, q):
    """B = 1 on two ranks: rank 1 owns no item and must still take part in every collective."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geomloss_amd import SamplesLoss
        from geomloss_amd.distributed import ShardedSamplesLoss, shard_batch

        torch.manual_seed(1)
        x, y = torch.rand(1, 30, 3), torch.rand(1, 35, 3)
        base = SamplesLoss("sinkhorn", p=2, blur=0.1, backend="tensorized")
        full = base(x, y)
        xl = shard_batch(x).clone().requires_grad_(True)
        yl = shard_batch(y)
        total = ShardedSamplesLoss(base, "sum")(xl, yl)
        (g,) = torch.autograd.grad(total, [xl])
        vec = ShardedSamplesLoss(base, "none")(xl.detach(), yl)
        mean = ShardedSamplesLoss(base, "mean")(xl.detach(), yl)
        ok_grad = g.shape == xl.shape and (rank == 1 or float(g.abs().sum()) > 0)
        q.put((rank, xl.shape[0], float((total - full.sum()).abs()), float((vec - full).abs().max()), float((mean - full.mean()).abs()),
               ok_grad, None))
    except Exception as e:
        q.put((rank, None, None, None, None, None, repr(e)))
    finally:
        dist.destroy_process_group()


def test_sharded_loss_with_a_rank_that_owns_nothing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_short_batch, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[1] for r in results) == [0, 1]
    for rank, _, terr, verr, merr, ok_grad, err in results:
        assert err is None, f"rank {rank}: {err}"
        assert terr < 1e-7 and verr < 1e-7 and merr < 1e-7 and ok_grad


def test_shard_bounds_partition_the_batch():
    from geomloss_amd.distributed import shard_bounds
    for B in (1, 7, 8, 256):
        for W in (1, 2, 4, 8):
            cuts = [shard_bounds(B, r, W) for r in range(W)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1


# ---- bench.py launches its own ranks (`python bench.py --gpus N` outside torchrun) -----------------------------------------

def _run_bench(*args, env=None, timeout=300):
    import subprocess
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                          env=e, cwd=ROOT)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2 --dry-run` with WORLD_SIZE unset: bench.py re-launches itself under torch.distributed.run
    (2 ranks, 127.0.0.1 rendezvous), the ranks meet over gloo, rank 0 prints ONE line with n_gpus = the ranks that really ran."""
    import json
    out = _run_bench("--gpus", "2", "--dry-run", "--steps", "3")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["dry_run"] is True and rec["steps"] == 3
    assert rec["items_all_ranks"] == 256.0          # the two shards of the default batch add up


def test_bench_refuses_instead_of_falling_back():
    """No silent single-GPU headline under `--gpus N`: a WORLD_SIZE that disagrees, or fewer GPUs than ranks, ends with a
    non-zero exit code and nothing on stdout."""
    out = _run_bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1"})
    assert out.returncode == 2 and out.stdout.strip() == "" and "refused" in out.stderr
    if not torch.cuda.is_available() or torch.cuda.device_count() < 64:
        out = _run_bench("--gpus", "64")
        assert out.returncode == 3 and out.stdout.strip() == "" and "refused" in out.stderr
