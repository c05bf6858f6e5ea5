"""Prioritized experience replay wired end to end on the device -- an EXTENSION.

The reference ships ``PrioritizedSamplingDistribution`` but cannot use it for learning: ``ReplayBuffer.sample``
drops the sampled keys (``replay_buffer.py:222-230``), ``collect_single_sample`` passes no priority
(``slimdqn/sample_collection/utils.py:27-35``) and the loss has no importance weights (``idqn.py:111-112``).
There is therefore no reference behaviour to match here; the pieces are checked against the oracle's weighted loss /
TD errors and against numpy restatements of the formulas (parity unpinned, by construction).

Design (Schaul et al. 2016, proportional variant), all on the GPU, no host synchronisation in the loop:

* ``SlotPrioritizedSampler``: the sum-tree leaf of an element IS its replay slot (``key % max_capacity``), so FIFO
  eviction is the overwrite of that leaf by the newcomer and no key <-> index map is needed on either side;
  newcomers enter with the running maximum priority, which lives in device memory.
* ``PrioritizedLearner.step()``: host draws B uniforms (PCG64) -> ``per_sample_leaves`` (tree descent, stratified)
  -> ``per_importance_weights`` -> ``replay_gather_stacked`` -> ``idqn_learn_on_batch`` with the weights, which also
  emits |TD| per head and sample -> ``per_priorities_from_td`` (mean or max over the K heads, ``(.+eps)^alpha``)
  -> ``sumtree_set`` on the same leaves.
"""
import numpy as np

from slimdqn import _hip
from slimdqn.sample_collection import sum_tree


class SlotPrioritizedSampler:
    """Sampler protocol of ``ReplayBuffer`` (add / remove / update / sample) with tree index == replay slot."""

    def __init__(self, seed: int, max_capacity: int, priority_exponent: float = 0.6):
        import torch

        self._max_capacity, self._alpha = max_capacity, priority_exponent
        self._sum_tree = sum_tree.SumTree(max_capacity)
        self._rng_key = np.random.default_rng(seed)
        self._key_of_slot = np.full(max_capacity, -1, np.int64)
        self._max_priority_dev = torch.ones(1, dtype=torch.float64, device="cuda")  # max_recorded_priority starts at 1
        self._size = 0

    def _set_one(self, slot: int, value: float = 0.0, value_dev=None) -> None:
        t = self._sum_tree
        _hip.check(_hip.lib().sumtree_set_one(_hip.ptr(t._nodes_dev), t._depth, int(slot), float(value), _hip.ptr(value_dev),
                                              _hip.current_stream()), "sumtree_set_one")

    def add(self, key, priority=None) -> None:
        slot = int(key) % self._max_capacity
        if self._key_of_slot[slot] < 0:
            self._size += 1
        self._key_of_slot[slot] = int(key)
        if priority is None:
            self._set_one(slot, value_dev=self._max_priority_dev)  # newcomers are sampled at least once soon
        else:
            self._set_one(slot, 0.0 if priority == 0.0 else float(priority) ** self._alpha)

    def remove(self, key) -> None:
        slot = int(key) % self._max_capacity
        if self._key_of_slot[slot] == int(key):  # not yet overwritten by its successor in the FIFO
            self._key_of_slot[slot] = -1
            self._size -= 1
            self._set_one(slot, 0.0)

    def update(self, keys, priorities) -> None:
        keys = np.atleast_1d(np.asarray(keys, np.int64))
        pr = np.atleast_1d(np.asarray(priorities, np.float64))
        self._sum_tree.set((keys % self._max_capacity).astype(np.int32), np.where(pr == 0.0, 0.0, pr**self._alpha))

    def sample(self, size: int):
        """Host-visible variant (synchronises): keys of ``size`` elements drawn in proportion to priority."""
        root = self._sum_tree.root
        slots = self._sum_tree.query(self._rng_key.uniform(0.0, root, size=size))
        return self._key_of_slot[slots].astype(np.int32)

    def __len__(self) -> int:
        return self._size


class PrioritizedLearner:
    """sample -> weights -> gather -> learn -> priorities -> tree update, queued on the stream without a host sync."""

    def __init__(self, agent, replay_buffer, beta: float = 0.4, eps: float = 1e-6, reduce: str = "mean",
                 stratified: bool = True):
        import torch

        assert isinstance(replay_buffer._sampling_distribution, SlotPrioritizedSampler)
        assert reduce in ("mean", "max")
        self.agent, self.rb, self.sampler = agent, replay_buffer, replay_buffer._sampling_distribution
        self.beta, self.eps, self.reduce_max, self.stratified = beta, eps, int(reduce == "max"), int(stratified)
        B, K = replay_buffer._batch_size, agent._K
        self._u_pin = torch.empty(B, dtype=torch.float64).pin_memory()
        self._u_ev = None
        self._u_dev = torch.empty(B, dtype=torch.float64, device="cuda")
        self._leaves = torch.empty(B, dtype=torch.int32, device="cuda")
        self._weights = torch.empty(B, dtype=torch.float32, device="cuda")
        self._td_abs = torch.zeros((K, B), dtype=torch.float32, device="cuda")
        self._priorities = torch.empty(B, dtype=torch.float64, device="cuda")

    def step(self):
        """One prioritized gradient step; returns the per-head losses (device tensor, not synchronised)."""
        import torch

        lib, q = _hip.lib(), _hip.current_stream()
        rb, tree, agent = self.rb, self.sampler._sum_tree, self.agent
        B = rb._batch_size
        assert rb.add_count, "No samples in replay buffer!"
        if self._u_ev is not None:
            self._u_ev.synchronize()  # the previous step's upload has left the pinned staging buffer
        self._u_pin.copy_(torch.from_numpy(self.sampler._rng_key.random(B)))
        self._u_dev.copy_(self._u_pin, non_blocking=True)
        self._u_ev = torch.cuda.Event()
        self._u_ev.record()
        _hip.check(lib.per_sample_leaves(_hip.ptr(tree._nodes_dev), tree._depth, _hip.ptr(self._u_dev), B, self.stratified,
                                         _hip.ptr(self._leaves), q), "per_sample_leaves")
        _hip.check(lib.per_importance_weights(_hip.ptr(tree._nodes_dev), tree._depth, _hip.ptr(self._leaves), B,
                                              len(self.sampler), self.beta, _hip.ptr(self._weights), q),
                   "per_importance_weights")
        batch = rb._gather_device(self._leaves)
        agent._ensure_handle(B)
        _hip.check(lib.idqn_set_per_buffers(agent._handle, _hip.ptr(self._weights), _hip.ptr(self._td_abs)),
                   "idqn_set_per_buffers")
        try:
            losses = agent._learn(batch)
        finally:
            _hip.check(lib.idqn_set_per_buffers(agent._handle, None, None), "idqn_set_per_buffers")
        _hip.check(lib.per_priorities_from_td(_hip.ptr(self._td_abs), agent._K, B, self.reduce_max, self.eps,
                                              self.sampler._alpha, _hip.ptr(self._priorities),
                                              _hip.ptr(self.sampler._max_priority_dev), q), "per_priorities_from_td")
        _hip.check(lib.sumtree_set(_hip.ptr(tree._nodes_dev), tree._depth, _hip.ptr(self._leaves),
                                   _hip.ptr(self._priorities), B, _hip.ptr(tree._scratch), q), "sumtree_set")
        return losses
