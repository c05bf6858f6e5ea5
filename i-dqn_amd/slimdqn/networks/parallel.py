"""Data-parallel i-DQN step over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-device; the loss is a plain mean over the minibatch (``slimdqn/networks/idqn.py:111-112``),
so the gradient of a global batch of ``B * world`` samples is the SUM of the shard gradients when every shard
divides by the global batch size.  The gradients are all-reduced, then the identical Adam update runs on every
rank, which keeps the replicas bit-identical.  ``torch.distributed`` is plumbing here: backend "nccl" is RCCL on
ROCm; the CPU tests drive the same function over "gloo".

Overlap (GPU, cnn path): 98 % of the 80.9 MB of gradients is Dense_0/kernel, and the backward pass produces it
FIRST.  The step is therefore issued as two C calls: ``idqn_learn_on_batch(..., IDQN_F_STOP_AFTER_DENSE0)`` queues
everything up to the Dense_0 weight gradient, the K contiguous Dense_0 slices are all-reduced asynchronously (RCCL
runs them on its own stream, ordered behind that point of the compute stream), and ``idqn_backward_rest`` queues
the conv backward, which then runs concurrently with the collective.  The gradient arena is laid out as two
contiguous regions for exactly this: ``[all heads' small leaves + the K losses][all heads' Dense_0 slices]`` -- two
collectives per step, no packing.  xGMI is point-to-point (7 links per GPU), so a few large collectives beat many
small ones.  ``IDQN_DP_OVERLAP=0`` falls back to one all-reduce of the whole arena after the backward pass.

Factored mode (GPU, cnn path, the default): the Dense_0/kernel gradient is the outer product ``a3^T . dh`` over the
samples, i.e. of rank <= global batch, far below its 7744 x 512 shape.  Ranks therefore ALL-GATHER the two factors
(``a3``: K x 7744 x 32 floats, ``dh``: K x 512 x 32 floats -- 5.3 MB per rank at K=5, one contiguous run inside the
library: nothing is copied before the collective) instead of all-reducing the
79.3 MB product, and every rank runs the fused weight-gradient + Adam kernel over the gathered global batch: 4x
(8 ranks) to 15x (2 ranks) fewer bytes over xGMI, the gradient is never materialised, and Adam stays fused.  The
gather runs while ``idqn_backward_rest`` computes the conv backward; the 1.6 MB of small leaves (and the K losses)
are all-reduced under the fused Dense_0 update, which needs only the gathered factors.  Every rank sums the same blocks in the same order, so replicas stay bit-identical.
``IDQN_DP_MODE=allreduce`` selects the all-reduce variants above.

Native mode (GPU, cnn path, RCCL backend: the default there): the SAME factored schedule issued by ONE C call,
``idqn_dp_step`` (``include/idqn_hip.h``, ``csrc/dp.hip``) -- the library owns an RCCL communicator (``idqn_dp_create``: rank 0's
``ncclGetUniqueId`` bytes are broadcast over this process group once) and enqueues forward -> ``ncclAllGather`` of the factor
run -> conv backward -> ``ncclAllReduce`` of the small region -> fused update itself, with the collectives on a side stream of
its own ordered by events (``IDQN_DP_STREAMS=inline``: on the compute stream).  The Python schedule below stays as the
oracle of that call (``tests/test_gpu_configs.py``: bit-identical) and as the path of the gloo tests.
``IDQN_DP_MODE=factored`` selects it on RCCL too.
"""
import ctypes as C
import os

import torch.distributed as dist

from slimdqn import _hip


def _factored_step(agent, shard, global_batch, group, extra_flags, serial=False):
    import torch

    lib, q = _hip.lib(), _hip.current_stream
    world = dist.get_world_size(group)
    K = agent._K
    F, J = next(shape for name, _, shape in agent._leaves if name == "Dense_0/kernel")
    nb = -(-len(shard.action) // 32)  # 32-sample blocks of this rank's shard
    X, Y = F * 32, J * 32
    n_a3, n_dh = K * nb * X, K * nb * Y
    key = (world, nb)
    if getattr(agent, "_factor_key", None) != key:  # gathered factors, allocated once per (world, shard blocks)
        agent._fact_all = torch.empty(world * (n_a3 + n_dh), dtype=torch.float32, device=agent._grad.device)  # [rank][dh | a3]
        agent._factor_key, agent._fact_send = key, None
    gathered = agent._fact_all
    agent._learn(shard, flags=_hip.F_STOP_BEFORE_DENSE0_WGRAD | extra_flags, mean_divisor=global_batch)
    # this rank's factors: dL/dh directly in front of the online nets' a3 inside the library -- one contiguous run, no copy
    p, c_dh, c_a3 = C.c_void_p(), C.c_int64(), C.c_int64()
    _hip.check(lib.idqn_dense0_factors(agent._handle, C.byref(p), C.byref(c_dh), C.byref(c_a3)), "idqn_dense0_factors")
    assert (c_dh.value, c_a3.value) == (n_dh, n_a3)
    if agent._fact_send is None or agent._fact_send.data_ptr() != p.value:
        agent._fact_send = _hip.device_view(p.value, n_dh + n_a3)
    send = agent._fact_send
    work = dist.all_gather_into_tensor(gathered, send, group=group, async_op=True)  # ONE collective for both factors
    if serial:  # test mode: no collective overlaps any kernel (the stream-race stress compares the two bit for bit)
        work.wait()
        torch.cuda.synchronize()
    _hip.check(lib.idqn_backward_rest(agent._handle, q()), "idqn_backward_rest")  # conv backward, under the gather
    if serial:
        torch.cuda.synchronize()
    small = dist.all_reduce(agent._grad_small, op=dist.ReduceOp.SUM, group=group, async_op=True)  # small leaves + K losses
    if serial:
        small.wait()
        torch.cuda.synchronize()
    work.wait()
    args = (agent._handle, _hip.ptr(gathered[n_dh:]), _hip.ptr(gathered), world * nb, nb, n_a3 + n_dh, nb * X, X,
            n_a3 + n_dh, nb * Y, Y)
    # the 80 us Dense_0 update needs only the gather: it runs while the small all-reduce is still in flight
    _hip.check(lib.idqn_finish_step_factored(*args, _hip.FACTORED_DENSE0, q()), "idqn_finish_step_factored")
    small.wait()
    _hip.check(lib.idqn_finish_step_factored(*args, _hip.FACTORED_REST, q()), "idqn_finish_step_factored")
    return agent._losses


def _native_handle(agent, group, streams=None):
    """The library-side RCCL communicator of this agent's handle (``idqn_dp_create``), created on first use: collective."""
    import torch

    dp = agent.__dict__.get("_dp")
    if dp is not None and dp[1] == agent._handle.value:
        return dp[0]
    if dp is not None:  # the handle was re-created for a larger batch: its dp object went with it (DeviceAgent._destroy_handle)
        agent.__dict__.pop("_dp", None)
    lib = _hip.lib()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    uid = (C.c_ubyte * _hip.DP_UNIQUE_ID_BYTES)()
    if rank == 0:
        _hip.check(lib.idqn_dp_unique_id(uid), "idqn_dp_unique_id")
    box = [bytes(uid)]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if streams is None:
        streams = os.environ.get("IDQN_DP_STREAMS", "side")
    flags = _hip.DP_SIDE_STREAM if streams == "side" else 0
    h = C.c_void_p()
    torch.cuda.synchronize()
    rc = lib.idqn_dp_create(agent._handle, box[0], rank, world, flags, C.byref(h))
    # every rank has to agree on the outcome: a communicator that came up on some ranks only would hang the first collective
    ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=agent._grad.device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        msg = lib.idqn_last_error().decode(errors="replace") if rc else "another rank failed"
        if rc == 0:
            lib.idqn_dp_destroy(h)
        raise _NativeUnavailable(f"idqn_dp_create: {msg}")
    agent._dp = (h, agent._handle.value)
    return h


class _NativeUnavailable(RuntimeError):
    """The library-side RCCL communicator could not be created on every rank (the Python schedule over torch.distributed runs instead)."""


def _native_step(agent, shard, global_batch, group, extra_flags, streams=None):
    B, ptrs = agent._prepare(shard)
    dp = _native_handle(agent, group, streams)
    _hip.check(_hip.lib().idqn_dp_step(dp, *ptrs, B, int(global_batch), int(extra_flags), _hip.current_stream()), "idqn_dp_step")
    return agent._losses


def data_parallel_step(agent, shard, global_batch: int, group=None, extra_flags: int = 0, overlap=None, mode=None,
                       serial: bool = False, streams=None):
    """One global gradient step; ``shard`` is this rank's ReplayElement-like slice of the global batch."""
    on_gpu_cnn = agent._grad.is_cuda and getattr(agent, "_arch", "") == "cnn"
    if mode is None:
        rccl = on_gpu_cnn and dist.get_backend(group) == "nccl"  # one rank per GPU: the library's own communicator works
        mode = os.environ.get("IDQN_DP_MODE", "native" if rccl else "factored")
    if mode == "native" and on_gpu_cnn and overlap is None and not serial and not getattr(agent, "_native_dp_failed", False):
        try:
            return _native_step(agent, shard, global_batch, group, extra_flags, streams)
        except _NativeUnavailable as e:  # agreed on by all ranks (see _native_handle): the same HIP kernels, collectives issued from Python
            import sys

            agent._native_dp_failed = True
            print(f"[idqn] native data-parallel step unavailable ({e}); using the Python schedule over torch.distributed", file=sys.stderr, flush=True)
    if mode == "native":
        mode = "factored"
    if mode == "factored" and on_gpu_cnn and overlap is None:
        return _factored_step(agent, shard, global_batch, group, extra_flags, serial)
    if overlap is None:
        overlap = (os.environ.get("IDQN_DP_OVERLAP", "1") != "0" and agent._grad.is_cuda
                   and getattr(agent, "_arch", "") == "cnn")
    if not overlap:
        agent._learn(shard, flags=_hip.F_GRADS_ONLY | extra_flags, mean_divisor=global_batch)
        dist.all_reduce(agent._grad, op=dist.ReduceOp.SUM, group=group)  # the whole arena in one collective
        if not getattr(agent, "_losses_in_grad", False):  # the device agent keeps its losses inside the arena
            dist.all_reduce(agent._losses, op=dist.ReduceOp.SUM, group=group)
        agent._apply_adam()  # Adam from the summed gradient, count += 1, cumulated_losses += losses
        return agent._losses
    # forward, head, Dense_0 data + weight gradient are queued; the conv backward is not yet
    agent._learn(shard, flags=_hip.F_STOP_AFTER_DENSE0 | extra_flags, mean_divisor=global_batch)
    big = dist.all_reduce(agent._grad_w0, op=dist.ReduceOp.SUM, group=group, async_op=True)  # 79.3 MB, 98 %
    if serial:
        import torch

        big.wait()
        torch.cuda.synchronize()
    _hip.check(_hip.lib().idqn_backward_rest(agent._handle, _hip.current_stream()), "idqn_backward_rest")
    dist.all_reduce(agent._grad_small, op=dist.ReduceOp.SUM, group=group)  # 1.6 MB of small leaves + the K losses
    big.wait()  # the compute stream waits for the Dense_0 region
    agent._apply_adam()
    return agent._losses


def shard_of(batch, rank: int, world: int):
    """Contiguous shard ``rank`` of ``world`` of every field of a ReplayElement-like batch."""
    n = len(batch.action)
    assert n % world == 0, f"global batch {n} is not divisible by {world} ranks"
    lo, hi = rank * (n // world), (rank + 1) * (n // world)
    return type(batch)(*[getattr(f, "tensor", f)[lo:hi] for f in batch])
