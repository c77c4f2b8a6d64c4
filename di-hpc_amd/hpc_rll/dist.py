"""Batch-axis data parallelism for the scalar-loss ops: one process per GPU, each rank holds a
contiguous (T, B/R) shard of the trajectories, every op computes per-rank SUMS scaled by 1/(GLOBAL count) and the
loss scalars (1 for TD-lambda/UPGO/TD family, 3 for V-trace, 5 for PPO) are summed with ONE all-reduce -- RCCL over
xGMI on GPUs (torch.distributed backend "nccl"), gloo in the CPU tests.  Per-sample outputs (adv, td_err) stay
sharded; GAE needs no collective at all.  The reference has no distributed path (no NCCL/MPI call sites).

The backward pass needs no communication: each rank's unit gradients already carry the global 1/count, so the
gradients equal the corresponding rows of the single-GPU result.
"""
from typing import Optional

import torch


def world_size(group=None) -> int:
    import torch.distributed as dist
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(group)


def loss_scale(local_count: int, group=None, sharded: bool = False) -> float:
    """1 / (global element count), assuming equal shards on every rank when ``sharded``."""
    n = max(int(local_count), 1)
    if sharded:
        n *= world_size(group)
    return 1.0 / n


def all_reduce_losses_(losses: torch.Tensor, group=None, sharded: bool = False, mean_slots=()) -> torch.Tensor:
    """In-place sum of the per-rank loss scalars.  ``mean_slots``: indices that hold per-rank MEANS (PPO's
    approx_kl / clipfrac monitors) and must be averaged instead of summed."""
    if not sharded:
        return losses
    import torch.distributed as dist
    ws = world_size(group)
    if ws == 1:
        return losses
    if mean_slots:
        idx = torch.as_tensor(list(mean_slots), device=losses.device)
        losses[idx] = losses[idx] / ws
    dist.all_reduce(losses, op=dist.ReduceOp.SUM, group=group)
    return losses


class _AllReduceSum(torch.autograd.Function):
    """Differentiable sum over the ranks of a few loss scalars: forward packs them into ONE buffer and all-reduces it,
    backward is the identity (d global / d local contribution = 1; every rank backpropagates its own shard)."""

    @staticmethod
    def forward(ctx, group, mean_slots, *losses):
        ctx.sizes = [t.numel() for t in losses]
        packed = torch.cat([t.reshape(-1) for t in losses])
        all_reduce_losses_(packed, group, True, mean_slots)
        # independent tensors (not views of `packed`): the caller may modify a returned loss in place
        return tuple(p.clone().view_as(t) for p, t in zip(torch.split(packed, ctx.sizes), losses))

    @staticmethod
    def backward(ctx, *grads):
        return (None, None) + grads


def all_reduce_sum(losses, group=None, mean_slots=()):
    """Sum the per-rank contributions of several loss tensors over the ranks with ONE all-reduce, keeping autograd
    intact (used by VTrace / PPO, whose three / five scalars are separate autograd outputs).  ``mean_slots`` index the
    flattened concatenation of ``losses``.  Returns new tensors; identity for a single rank."""
    if world_size(group) == 1:
        return tuple(losses)
    return _AllReduceSum.apply(group, tuple(mean_slots), *losses)


def all_reduce_max_int(value: int, group=None, device=None) -> int:
    """max over the ranks of a python int (a globally consistent pad width / max shape for the
    batch-sharded Pad ops).  One all-reduce(MAX) of one int64; identity for a single rank."""
    import torch.distributed as dist
    if world_size(group) == 1:
        return int(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def shard_batch(t: Optional[torch.Tensor], dim: int, rank: int, world: int) -> Optional[torch.Tensor]:
    """Contiguous shard ``rank`` of ``world`` along the batch axis ``dim`` (None passes through)."""
    if t is None:
        return None
    n = t.shape[dim]
    assert n % world == 0, f"batch {n} not divisible by world size {world}"
    k = n // world
    return t.narrow(dim, rank * k, k).contiguous()


def all_gather_batch(t: torch.Tensor, dim: int, group=None) -> torch.Tensor:
    """Replicate a batch-sharded per-sample output (``adv``, ``td_err``, ...) on every rank: the (.., B/R, ..)
    shards are gathered to (R, ...) with ONE all_gather (RCCL over xGMI: the first place the 7 x ~153 GB/s link budget
    matters) and re-interleaved along ``dim`` so the result equals the unsharded tensor.  No gradient flows through it."""
    import torch.distributed as dist
    ws = world_size(group)
    if ws == 1:
        return t
    t = t.contiguous()
    flat = torch.empty(ws * t.numel(), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(flat, t.reshape(-1), group=group)       # concatenation form: every backend has it
    out = flat.view((ws,) + tuple(t.shape))
    # (R, d0, .., B/R, ..) -> (d0, .., R, B/R, ..) -> (d0, .., B, ..)
    out = out.movedim(0, dim)
    shape = list(t.shape)
    shape[dim] *= ws
    return out.reshape(shape)


def all_reduce_grads_(params, group=None, bucket_bytes: int = 256 << 20, average: bool = False) -> None:
    """Sum (or average) the ``.grad`` of ``params`` over the ranks, in place -- the one exchange step of the
    batch-sharded LSTM (dWx, dWh, dbias, dgamma, dbeta; activations never travel).

    The gradients are packed into flat buckets and each bucket is ONE all-reduce: RCCL rings over xGMI are
    per-link bound (7 x ~153 GB/s) and pay a fixed latency per collective, so a few large messages beat one
    collective per parameter.  The default bucket (256 MiB) takes the whole C4 LSTM (34 MB of weight
    gradients) in a single collective; HBM is not the constraint at 288 GB."""
    import torch.distributed as dist
    ws = world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    if ws == 1 or not grads:
        return
    buckets, cur, cur_bytes = [], [], 0
    for g in grads:
        nbytes = g.numel() * g.element_size()
        if cur and (cur_bytes + nbytes > bucket_bytes or g.dtype != cur[0].dtype or g.device != cur[0].device):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(g)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    for bucket in buckets:
        if len(bucket) == 1 and bucket[0].is_contiguous():
            flat = bucket[0].view(-1)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                flat.div_(ws)
            continue
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(ws)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
