"""Head-parallel i-DQN over the GPUs of one node (BASELINE config 5): K heads, K / world consecutive
heads per rank, one process per GPU.

The K heads of i-DQN are independent inside a gradient step (``idqn.py:96-109`` vmaps over them), so with the same
minibatch on every rank (same-seed sampling) a step needs NO collective at all.  Only the chain maintenance crosses
rank boundaries, and only between direct neighbours (one xGMI link each, 16.2 MB per message for the Nature-CNN):

* T-step (``idqn.py:78-80``): ``target <- online``; ``online[k] <- online[k+1]``.  The last local head of rank g
  takes the OLD first head of rank g+1; the global last head keeps its value.
* D-step (``idqn.py:20-24,92``): ``target[k] <- online[k-1]`` for k >= 1.  The first local head of rank g > 0 takes
  the last head of rank g-1; global head 0 keeps its target.

``torch.distributed`` point-to-point ops (RCCL send / recv on "nccl", plain sockets on "gloo" in the CPU tests) carry
the rows; everything else is the single-device agent on a K / world window of the heads.
"""
import numpy as np

from slimdqn import prng
from slimdqn.networks.idqn import iDQN


def head_window(n_networks: int, rank: int, world: int):
    """(first, count) of the consecutive heads rank ``rank`` owns."""
    assert n_networks % world == 0, f"{n_networks} heads do not divide over {world} ranks"
    count = n_networks // world
    return rank * count, count


def _exchange(sends, recvs, group=None):
    """Posts the point-to-point operations ``[(tensor, peer)]`` and returns a function that completes them.  RCCL
    moves device rows directly; gloo (CPU tests, or two test ranks sharing one GPU) cannot receive into device memory,
    so device rows are staged through host copies there."""
    import torch.distributed as dist

    staged = dist.get_backend(group) == "gloo"
    ops, copies = [], []
    for t, peer in sends:
        ops.append(dist.P2POp(dist.isend, t.cpu() if (staged and t.is_cuda) else t, peer, group))
    for t, peer in recvs:
        if staged and t.is_cuda:
            host = t.new_empty(t.shape, device="cpu")
            copies.append((t, host))
            ops.append(dist.P2POp(dist.irecv, host, peer, group))
        else:
            ops.append(dist.P2POp(dist.irecv, t, peer, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []

    def finish():
        for r in reqs:
            r.wait()
        for dev, host in copies:
            dev.copy_(host)

    return finish


def sharded_target_update(agent, rank: int, world: int, group=None) -> None:
    """The T-step over a head-sharded chain; ``agent`` holds the rows ``_online`` / ``_target`` of its window."""
    sends, recvs, incoming = [], [], None
    if rank > 0:  # my OLD first head becomes the new last head of the rank below
        sends.append((agent._online[0].clone(), rank - 1))
    if rank < world - 1:
        incoming = agent._online.new_empty(agent._online.shape[1])
        recvs.append((incoming, rank + 1))
    finish = _exchange(sends, recvs, group)
    agent._local_target_update()  # target <- online, shift inside the window (its last head keeps its value)
    finish()
    if incoming is not None:
        agent._online[-1].copy_(incoming)


def sharded_target_sync(agent, rank: int, world: int, group=None) -> None:
    """The D-step over a head-sharded chain (neither side of the exchange is touched by the local sync)."""
    sends = [(agent._online[-1], rank + 1)] if rank < world - 1 else []
    recvs = [(agent._target[0], rank - 1)] if rank > 0 else []
    finish = _exchange(sends, recvs, group)
    agent._local_target_sync()  # target[j] <- online[j-1] for the local j >= 1
    finish()


class HeadShardedIDQN(iDQN):
    """``iDQN`` whose ``n_networks`` heads are spread over the ranks of ``group`` (same constructor otherwise).

    ``params`` / ``target_params`` / ``optimizer_state`` expose this rank's window (leading axis K / world);
    ``n_networks`` stays the global K, the log keys of ``update_target_params`` cover all K heads, and
    ``best_action`` returns the same action on every rank.
    """

    def __init__(self, key, observation_dim, n_actions, n_networks, features, architecture_type, learning_rate, gamma,
                 update_horizon, update_to_data, target_update_frequency, target_sync_frequency, adam_eps=1e-8,
                 group=None):
        import torch.distributed as dist

        self._group = group
        self._rank, self._world = dist.get_rank(group), dist.get_world_size(group)
        self._head_first, self._n_local = head_window(n_networks, self._rank, self._world)
        super().__init__(key, observation_dim, n_actions, n_networks, features, architecture_type, learning_rate, gamma,
                         update_horizon, update_to_data, target_update_frequency, target_sync_frequency, adam_eps,
                         _local_heads=(self._head_first, self._n_local))

    def _target_update(self) -> None:
        sharded_target_update(self, self._rank, self._world, self._group)

    def _target_sync(self) -> None:
        sharded_target_sync(self, self._rank, self._world, self._group)

    def _all_cumulated_losses(self) -> np.ndarray:
        import torch
        import torch.distributed as dist

        out = torch.empty(self.n_networks, dtype=torch.float64, device=self._cum.device)
        dist.all_gather_into_tensor(out, self._cum, group=self._group)
        return out.cpu().numpy()

    def best_action(self, params, state, key):
        """The head is drawn from the key over ALL K heads (idqn.py:128); its owner evaluates, everyone gets the action."""
        import torch
        import torch.distributed as dist

        idx = prng.randint(key, 0, self.n_networks)
        owner, local = divmod(idx, self._n_local)
        action = torch.zeros(1, dtype=torch.int64, device=self._cum.device)
        if owner == self._rank:
            assert params is self.params or params is self.target_params
            action[0] = int(self._best_action(0 if params is self.params else 1, local, state))  # (also a pending host action)
        dist.broadcast(action, src=dist.get_global_rank(self._group, owner) if self._group is not None else owner,
                       group=self._group)
        return action[0]

    def get_model(self):
        """This rank's window, tagged with its place in the chain."""
        model = super().get_model()
        model["heads"] = (self._head_first, self._n_local, self.n_networks)
        return model
