"""CPU, world_size 2 and 4 over gloo: the head-parallel chain maintenance (slimdqn/networks/head_parallel.py).

Each rank holds a window of the K heads as plain CPU tensors and runs the same neighbour exchanges the GPU agent
runs over RCCL; after an arbitrary sequence of T-steps (copy + shift, idqn.py:78-80) and D-steps (sync,
idqn.py:20-24) interleaved with "learning" (head-specific perturbations of the online rows), the windows put side by
side must equal the oracle's shift_params / sync_target_params applied to the unsharded K-head arrays -- bit for bit,
these are copies.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

K, P = 8, 37
SCHEDULE = "LDLDLTLDDLTTLDLTD"  # L: learn, D: sync step, T: target update step


class WindowAgent:
    """The attributes sharded_target_update / sharded_target_sync touch, on the CPU."""

    def __init__(self, online, target):
        self._online, self._target = online.clone(), target.clone()

    def _local_target_update(self):
        self._target.copy_(self._online)
        self._online[:-1] = self._online[1:].clone()

    def _local_target_sync(self):
        self._target[1:] = self._online[:-1]


def _initial():
    g = torch.Generator().manual_seed(5)
    online = torch.rand(K, P, generator=g, dtype=torch.float32)
    return online, online.clone()


def _perturb(online, first, step):
    """What a gradient step does to the chain as far as this test cares: every head moves, differently."""
    for j in range(online.shape[0]):
        online[j] += 0.001 * (first + j + 1) * (step + 1)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from slimdqn.networks.head_parallel import head_window, sharded_target_sync, sharded_target_update

        first, count = head_window(K, rank, world)
        online, target = _initial()
        agent = WindowAgent(online[first : first + count], target[first : first + count])
        for step, op in enumerate(SCHEDULE):
            if op == "L":
                _perturb(agent._online, first, step)
            elif op == "T":
                sharded_target_update(agent, rank, world)
            else:
                sharded_target_sync(agent, rank, world)
        torch.save({"online": agent._online, "target": agent._target, "first": first}, f"{out}/rank{rank}.pt")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_chain_equals_the_unsharded_chain(tmp_path, world):
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "i-dqn_amd"))
    from oracle import qnet_ref as Q

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    online, target = _initial()
    p, pt = {"w": online.numpy().copy()}, {"w": target.numpy().copy()}
    for step, op in enumerate(SCHEDULE):
        if op == "L":
            t = torch.from_numpy(p["w"])
            _perturb(t, 0, step)
        elif op == "T":  # idqn.py:78-80: target = params.copy(); params = shift_params(params)
            pt = {"w": p["w"].copy()}
            p = Q.shift_params(p)
        else:
            pt = Q.sync_target_params(p, pt)
    got_online = np.concatenate([torch.load(f"{tmp_path}/rank{r}.pt")["online"].numpy() for r in range(world)])
    got_target = np.concatenate([torch.load(f"{tmp_path}/rank{r}.pt")["target"].numpy() for r in range(world)])
    np.testing.assert_array_equal(got_online, p["w"])
    np.testing.assert_array_equal(got_target, pt["w"])


def test_head_window_partition():
    from slimdqn.networks.head_parallel import head_window

    assert [head_window(64, r, 8) for r in (0, 3, 7)] == [(0, 8), (24, 8), (56, 8)]
    with pytest.raises(AssertionError):
        head_window(5, 0, 2)
