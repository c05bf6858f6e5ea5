"""Data-parallel i-DQN step over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-device; the loss is a plain mean over the minibatch (``slimdqn/networks/idqn.py:111-112``),
so the gradient of a global batch of ``B * world`` samples is the SUM of the shard gradients when every shard
divides by the global batch size.  One all-reduce of the ``[K][head_stride]`` f32 gradient arena (80.9 MB at K=5)
plus the K per-head losses, then the identical Adam update on every rank keeps the replicas bit-identical.
``torch.distributed`` is plumbing here: backend "nccl" is RCCL on ROCm; the CPU tests drive the same function
over "gloo".
"""
import torch.distributed as dist

from slimdqn import _hip


def data_parallel_step(agent, shard, global_batch: int, group=None, extra_flags: int = 0):
    """One global gradient step; ``shard`` is this rank's ReplayElement-like slice of the global batch."""
    agent._learn(shard, flags=_hip.F_GRADS_ONLY | extra_flags, mean_divisor=global_batch)  # gradients + loss share
    dist.all_reduce(agent._grad, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(agent._losses, op=dist.ReduceOp.SUM, group=group)
    agent._apply_adam()  # Adam from the summed gradient, count += 1, cumulated_losses += losses
    return agent._losses


def shard_of(batch, rank: int, world: int):
    """Contiguous shard ``rank`` of ``world`` of every field of a ReplayElement-like batch."""
    n = len(batch.action)
    assert n % world == 0, f"global batch {n} is not divisible by {world} ranks"
    lo, hi = rank * (n // world), (rank + 1) * (n // world)
    return type(batch)(*[getattr(f, "tensor", f)[lo:hi] for f in batch])
