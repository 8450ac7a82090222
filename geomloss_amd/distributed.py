"""Batch-sharded ``SamplesLoss`` over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed code at all; this module is new.  The B problems of a batched loss
are independent (every tensor of the Sinkhorn loop carries the leading batch axis and no operation
mixes batch items), so the data path shards with *no* collective.  Two tiny collectives remain:

1. only when ``diameter`` is None: a MIN and a MAX all-reduce of the D bounding-box coordinates, because
   the reference derives one epsilon schedule from the bounding box of the *whole flattened batch*
   (``_legacy/sinkhorn_divergence.py:156-158``); every rank must run the same schedule as the unsharded
   computation would.
2. after the loss: a SUM all-reduce of one scalar (``reduction="sum"|"mean"``) or an all-gather of the
   (B/W,) local loss vectors (``reduction="none"``).

Both are a few bytes: latency-bound, nothing to overlap or bucket.  The backward pass needs no
communication (d loss_b / d x_b is local to the rank that owns item b).

Use ``backend="nccl"`` (= RCCL) on GPUs; the same code runs on ``gloo`` for CPU tests.
"""

import copy

import torch
import torch.distributed as dist


def shard_bounds(B, rank, world_size):
    """Contiguous batch slice [lo, hi) owned by ``rank``; the first B % W ranks get one extra item."""
    base, rem = divmod(B, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(t, rank=None, world_size=None):
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    lo, hi = shard_bounds(t.shape[0], rank, world_size)
    return t[lo:hi]


def global_diameter(x, y, group=None):
    """Bounding-box diagonal of the union of all ranks' (flattened) clouds: 2 all-reduces of D floats."""
    D = x.shape[-1]
    xf, yf = x.reshape(-1, D), y.reshape(-1, D)
    if xf.shape[0] == 0 or yf.shape[0] == 0:     # a rank without batch items (B < world size): neutral for MIN / MAX
        mins = torch.full((D,), float("inf"), device=x.device)
        maxs = torch.full((D,), float("-inf"), device=x.device)
    else:
        mins = torch.minimum(xf.min(dim=0)[0], yf.min(dim=0)[0]).float()
        maxs = torch.maximum(xf.max(dim=0)[0], yf.max(dim=0)[0]).float()
    dist.all_reduce(mins, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(maxs, op=dist.ReduceOp.MAX, group=group)
    return (maxs - mins).norm().item()


class ShardedSamplesLoss(torch.nn.Module):
    """Wraps a :class:`SamplesLoss`; each rank passes ITS shard of the batch.

    ``forward(α, x, β, y)`` or ``forward(x, y)`` with (B_local, N, D) clouds returns

    * ``reduction="sum"`` / ``"mean"``: the scalar loss over the *global* batch, identical on every rank;
      its gradient with respect to the local shard is the exact gradient of the global loss.
    * ``reduction="none"``: the (B_global,) vector of losses, rank-major (= original order when the
      batch was cut with :func:`shard_batch`); entries of other ranks carry no gradient.
    """

    def __init__(self, loss, reduction="sum", group=None):
        super().__init__()
        if reduction not in ("sum", "mean", "none"):
            raise ValueError("reduction should be 'sum', 'mean' or 'none'.")
        self.loss, self.reduction, self.group = loss, reduction, group

    def forward(self, *args):
        loss = self.loss
        x, y = (args[0], args[1]) if len(args) == 2 else (args[1], args[3])
        if x.dim() != 3:
            raise ValueError("ShardedSamplesLoss expects batched (B_local, N, D) clouds.")
        if getattr(loss, "diameter", None) is None and getattr(loss, "loss", None) == "sinkhorn":
            loss = copy.copy(loss)  # same schedule on every rank as the unsharded reference computation
            loss.diameter = global_diameter(x.detach(), y.detach(), self.group)
            loss._diameter_bounds_the_data = True       # measured: the HIP soft-mins may size their exponent layout on it
        if x.shape[0] == 0:      # nothing on this rank: it still takes part in the collectives above and below
            local = x.new_zeros(0, dtype=torch.float32) + 0.0 * x.sum()
        else:
            local = loss(*args)  # (B_local,)

        if self.reduction == "none":
            sizes = [torch.zeros(1, dtype=torch.long, device=local.device) for _ in range(dist.get_world_size(self.group))]
            dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.long, device=local.device), group=self.group)
            nmax = int(max(s.item() for s in sizes))
            pad = torch.zeros(nmax, dtype=local.dtype, device=local.device)
            pad[: local.shape[0]] = local.detach()
            parts = [torch.empty_like(pad) for _ in sizes]
            dist.all_gather(parts, pad, group=self.group)
            rank = dist.get_rank(self.group)
            out = [p[: int(s.item())] for p, s in zip(parts, sizes)]
            out[rank] = local  # keep the autograd graph of the local entries
            return torch.cat(out)

        total = local.sum()
        reduced = total.detach().clone()
        count = torch.tensor([float(local.shape[0])], device=local.device)
        dist.all_reduce(reduced, op=dist.ReduceOp.SUM, group=self.group)
        # value = global sum; gradient = gradient of the local sum (other ranks' terms do not depend on local inputs)
        out = total + (reduced - total.detach())
        if self.reduction == "mean":
            dist.all_reduce(count, op=dist.ReduceOp.SUM, group=self.group)
            out = out / count[0]
        return out
