"""CPU, world_size 2 over gloo: the data-parallel step's host logic (slimdqn/networks/parallel.py).

The device kernels cannot run here, so each rank drives ``data_parallel_step`` with an agent whose
``_learn`` / ``_apply_adam`` are the CPU oracle (same contract as the C ABI: GRADS_ONLY writes the shard's
gradient with the GLOBAL mean divisor into ``_grad``, ``_apply_adam`` applies Adam from ``_grad``).  After the
all-reduce both ranks must hold exactly the full-batch oracle step.
"""
import os
import socket
from collections import namedtuple

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

Batch = namedtuple("Batch", "state action reward next_state is_terminal")
ARCH, OBS, A, FEATS, K, GB = "fc", 8, 4, [24, 16], 3, 16
LR, EPS, GAMMA = 1e-3, 1e-8, 0.99


def _inputs():
    from oracle import qnet_ref as Q

    p = Q.init_params(0, ARCH, OBS, A, FEATS, K, np.float64)
    pt = Q.init_params(1, ARCH, OBS, A, FEATS, K, np.float64)
    s, a, r, s2, t = Q.synthetic_batch(2, GB, OBS, A, ARCH)
    t[3] = True
    return p, pt, Batch(s, a, r, s2, t)


class OracleAgent:
    """Stand-in with the attributes data_parallel_step touches."""

    def __init__(self, p, pt):
        from oracle import qnet_ref as Q

        self.Q, self.p, self.pt = Q, {n: v.copy() for n, v in p.items()}, pt
        self.names = list(p)
        self.sizes = [int(np.prod(p[n].shape[1:])) for n in self.names]
        self._grad = torch.zeros(K, sum(self.sizes), dtype=torch.float64)
        self._losses = torch.zeros(K, dtype=torch.float64)
        self.mu = {n: np.zeros_like(v) for n, v in p.items()}
        self.nu = {n: np.zeros_like(v) for n, v in p.items()}
        self.count = np.zeros(K, np.int64)

    def _learn(self, batch, flags=0, mean_divisor=None):
        assert flags & 1, "the data-parallel step must ask for gradients only"
        n = len(batch.action)
        for k in range(K):
            loss, grads, _ = self.Q.loss_and_grads(self.Q.head(self.p, k), self.Q.head(self.pt, k), tuple(batch), ARCH, GAMMA)
            scale = n / mean_divisor  # oracle averages over the shard; the contract averages over the global batch
            self._losses[k] = loss * scale
            self._grad[k] = torch.from_numpy(np.concatenate([grads[m].ravel() for m in self.names]) * scale)
        return self._losses

    def _apply_adam(self):
        g = self._grad.numpy()
        for k in range(K):
            off = 0
            for n, sz in zip(self.names, self.sizes):
                gk = g[k, off : off + sz].reshape(self.p[n].shape[1:])
                self.p[n][k], self.mu[n][k], self.nu[n][k] = self.Q.adam_update(
                    self.p[n][k], gk, self.mu[n][k], self.nu[n][k], self.count[k], LR, EPS)
                off += sz
        self.count += 1


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from slimdqn.networks.parallel import data_parallel_step, shard_of

        p, pt, batch = _inputs()
        agent = OracleAgent(p, pt)
        losses = data_parallel_step(agent, shard_of(batch, rank, world), GB)
        torch.save({"p": agent.p, "losses": losses.numpy().copy(), "count": agent.count}, f"{out}/rank{rank}.pt")
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_equals_full_batch_step(tmp_path):
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "i-dqn_amd"))
    from oracle import qnet_ref as Q

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p, pt, batch = _inputs()
    zeros = {n: np.zeros_like(v) for n, v in p.items()}
    want_p, _, _, _, want_l = Q.learn_on_batch(p, pt, zeros, zeros, np.zeros(K, np.int64), tuple(batch), ARCH, GAMMA, LR, EPS)
    for rank in range(2):
        got = torch.load(f"{tmp_path}/rank{rank}.pt", weights_only=False)
        np.testing.assert_allclose(got["losses"], want_l, rtol=1e-12)
        assert got["count"].tolist() == [1] * K
        for n in want_p:
            np.testing.assert_allclose(got["p"][n], want_p[n], rtol=1e-10, atol=1e-14)


def test_shard_of_partitions_the_batch():
    from slimdqn.networks.parallel import shard_of

    _, _, batch = _inputs()
    parts = [shard_of(batch, r, 4) for r in range(4)]
    np.testing.assert_array_equal(np.concatenate([p.action for p in parts]), batch.action)
    with pytest.raises(AssertionError):
        shard_of(batch, 0, 3)
