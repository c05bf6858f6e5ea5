"""Sampling distributions: host index maps + numpy PCG64 stream, priorities in the HBM sum tree.

Drop-in for the reference's ``slimdqn/sample_collection/samplers.py``.  What stays on the host is
exactly what is integer bookkeeping in the reference too: the dense ``index -> key`` array with
swap-remove (``samplers.py:26-37``) and the ``numpy.random.Generator`` whose stream defines the
sampled indices (``:17,43,110``) -- keeping numpy's generator IS bit-parity of the index stream.
Priorities and the inverse-CDF query run on the device (``sum_tree.SumTree``).  The prioritized sampler also keeps
``index -> key`` in HBM (written by the same launches that write the priorities), so that ``sample`` is ONE launch and
ONE host read (leaves -> keys on the device, root and status in the same mailbox) and ``remove`` reads the moved priority
on the device instead of the host.  The host dict / list stay as the mirror the reference's tests reach into and as the
argument check of ``remove`` (``samplers.py:27``); ``key -> index`` has no device consumer.  The uniform sampler is host-only
by default (a host generator feeding host arithmetic); ``enable_device_map`` gives it the same device copy of ``index -> key``
(``sampler_map_set`` per add / remove) and ``sample_device`` hands out the sampled keys as a device tensor.
"""
import numpy as np
import torch

from slimdqn import _hip

from slimdqn.sample_collection import ReplayItemID, sum_tree


class IndexMap:
    """Dense local index <-> replay key map with O(1) swap-remove (pure host logic, no GPU)."""

    def __init__(self) -> None:
        self.key_to_index = {}
        self.index_to_key = []

    def __len__(self) -> int:
        return len(self.index_to_key)

    def add(self, key) -> int:
        self.key_to_index[key] = len(self.index_to_key)
        self.index_to_key.append(key)
        return len(self.index_to_key) - 1

    def remove(self, key):
        """Returns (hole, last): the index freed and the index whose entry moved into it."""
        assert key in self.key_to_index, ValueError(f"Key {key} not found.")
        hole = self.key_to_index.pop(key)
        last = len(self.index_to_key) - 1
        moved = self.index_to_key.pop()
        if moved != key:
            self.index_to_key[hole] = moved
            self.key_to_index[moved] = hole
        return hole, last

    def keys_at(self, indices) -> np.ndarray:
        table = self.index_to_key
        return np.fromiter((table[i] for i in indices), dtype=np.int32, count=len(indices))


class UniformSamplingDistribution:
    """samplers.py:13-49."""

    def __init__(self, seed: int) -> None:
        self._rng_key = np.random.default_rng(seed)
        self._map = IndexMap()

    # attribute names the reference's tests reach into (tests/test_replay_buffer.py:161,246-273)
    _key_to_index = property(lambda self: self._map.key_to_index)
    _index_to_key = property(lambda self: self._map.index_to_key)

    _i2k_dev = None  # optional device copy of index -> key (enable_device_map)

    def enable_device_map(self, capacity: int) -> None:
        """Keeps a copy of ``index -> key`` in HBM (``sampler_map_set`` on every add / remove) so that ``sample_device``
        can hand out the sampled keys as a device tensor.  Not part of the reference's protocol: opt-in."""
        assert int(capacity) >= len(self._map), "capacity below the number of keys already held"
        self._i2k_dev = torch.zeros(int(capacity), dtype=torch.int32, device="cuda")
        for i, k in enumerate(self._map.index_to_key):
            _hip.check(_hip.lib().sampler_map_set(_hip.ptr(self._i2k_dev), i, int(k), _hip.current_stream()), "sampler_map_set")

    def add(self, key: ReplayItemID) -> None:
        if self._i2k_dev is not None and len(self._map) >= self._i2k_dev.numel():
            raise IndexError(f"device index map holds {self._i2k_dev.numel()} entries (enable_device_map capacity)")
        index = self._map.add(key)
        if self._i2k_dev is not None:
            _hip.check(_hip.lib().sampler_map_set(_hip.ptr(self._i2k_dev), index, int(key), _hip.current_stream()), "sampler_map_set")

    def remove(self, key: ReplayItemID) -> None:
        hole, last = self._map.remove(key)
        if self._i2k_dev is not None and hole != last:  # the last entry moved into the hole (samplers.py:31-35)
            _hip.check(_hip.lib().sampler_map_set(_hip.ptr(self._i2k_dev), hole, int(self._map.index_to_key[hole]),
                                                  _hip.current_stream()), "sampler_map_set")

    def sample(self, size: int):
        assert self._map.index_to_key, ValueError("No keys to sample from.")
        return self._map.keys_at(self._rng_key.integers(len(self._map), size=size))

    def sample_device(self, size: int):
        """``sample`` with the keys left on the device (int32 tensor): the same generator draw, mapped by ``sampler_map_indices``."""
        assert self._i2k_dev is not None, "enable_device_map first"
        assert self._map.index_to_key, ValueError("No keys to sample from.")
        idx = torch.from_numpy(self._rng_key.integers(len(self._map), size=size).astype(np.int32)).cuda()
        out = torch.empty(int(idx.numel()), dtype=torch.int32, device="cuda")
        _hip.check(_hip.lib().sampler_map_indices(_hip.ptr(self._i2k_dev), _hip.ptr(idx), int(idx.numel()), _hip.ptr(out),
                                                  _hip.current_stream()), "sampler_map_indices")
        return out


class PrioritizedSamplingDistribution(UniformSamplingDistribution):
    """samplers.py:52-116, with the sum tree in HBM."""

    def __init__(self, seed: int, max_capacity: int, priority_exponent: float = 1.0) -> None:
        self._max_capacity = max_capacity
        self._priority_exponent = priority_exponent
        self._sum_tree = sum_tree.SumTree(self._max_capacity)
        self._i2k_dev = torch.zeros(self._max_capacity, dtype=torch.int32, device="cuda")  # index -> key, device copy
        # add / remove / update / sample all launch on the tree's own stream: `sample` (one launch + a polled host mailbox,
        # samplers.py:105-116) then returns in ~15 us whatever the learner has queued on its stream, instead of waiting for
        # the gradient step in flight -- the same operations in the same order on the same tree, so the same keys
        self._sum_tree.use_own_stream()
        super().__init__(seed=seed)

    def add(self, key: ReplayItemID, priority: float) -> None:
        index = self._map.add(key)
        if priority is None:
            priority = 0.0
        value = 0.0 if priority == 0.0 else priority**self._priority_exponent
        tree = self._sum_tree
        assert value >= 0.0, "Values must be positive."
        if not 0 <= index < self._max_capacity:
            raise IndexError("sum tree index out of range")
        tree.max_recorded_priority = max(tree.max_recorded_priority, value)
        # one launch: index_to_key[index] = key and tree.set(index, value)   (samplers.py:62-66)
        _hip.check(_hip.lib().sampler_prioritized_add(_hip.ptr(tree._nodes_dev), tree._depth, _hip.ptr(self._i2k_dev), int(index),
                                                      int(key), float(value), tree._q()), "sampler_prioritized_add")

    def update(self, keys, priorities) -> None:
        if not isinstance(keys, np.ndarray):
            keys = np.asarray([keys], dtype=np.int32)
        shaped = np.where(priorities == 0.0, 0.0, priorities**self._priority_exponent)
        local = np.fromiter((self._map.key_to_index[k] for k in keys), dtype=np.int32)
        self._sum_tree.set(local, shaped)

    def remove(self, key: ReplayItemID) -> None:
        hole = self._map.key_to_index[key]  # unknown key: KeyError, as the reference's `self._key_to_index[key]` (samplers.py:90)
        last = len(self._map) - 1
        # one launch, no host read: the last entry's priority moves into the hole by the two-leaf set
        # {hole: leaf[last], last: 0} (samplers.py:98-102; hole == last: {hole: 0}), its key in the device map
        tree = self._sum_tree
        _hip.check(_hip.lib().sampler_prioritized_remove(_hip.ptr(tree._nodes_dev), tree._depth, _hip.ptr(self._i2k_dev), int(hole),
                                                         int(last), tree._q()), "sampler_prioritized_remove")
        self._map.remove(key)

    def sample(self, size: int):
        assert self._map.index_to_key, ValueError("No keys to sample from.")
        # `Generator.uniform(0.0, root, size)` is 0.0 + root * next_double per element (samplers.py:110): the host draws the
        # doubles (the PCG64 stream IS the parity), the device multiplies by the root it holds -- no read of the root first
        before = self._rng_key.bit_generator.state
        u = self._rng_key.random(size)
        _, keys, root, status = self._sum_tree.query_host(u, scale_by_root=True, index_to_key=self._i2k_dev, n_live=len(self._map))
        if root == 0.0:
            # the reference's branch here is `super().sample(size).keys` -> AttributeError (samplers.py:106-108), after
            # drawing `integers` from the generator: replay exactly that on the restored stream
            self._rng_key.bit_generator.state = before
            super().sample(size)
            raise AttributeError("'numpy.ndarray' object has no attribute 'keys'")
        if status & 1:
            raise ValueError(f"Targets must be in the interval [0.0, {root}).")
        assert not (status & 2), "sum tree traversal: target not below its node (sum_tree.py:81)"
        if status & 4:  # a descent ended on an empty leaf behind the live entries: `self._index_to_key[index]` (samplers.py:114)
            raise IndexError("list index out of range")
        return keys

    def enable_device_map(self, capacity: int) -> None:
        pass  # (always on: add / remove write the map in the launches that write the priorities)

    def sample_device(self, size: int):
        """``sample`` with the keys left on the device and no host read at all: the same generator draw, targets ``u * root``
        made on the device (clamped below the root where the reference would raise), leaves mapped by ``sampler_map_indices``."""
        assert self._map.index_to_key, ValueError("No keys to sample from.")
        tree = self._sum_tree
        tree._sync_to_current()  # (this variant runs on the caller's stream, behind everything the tree's stream has done)
        u = torch.from_numpy(self._rng_key.random(size)).cuda()
        leaves = torch.empty(int(u.numel()), dtype=torch.int32, device="cuda")
        keys = torch.empty_like(leaves)
        lib, q = _hip.lib(), _hip.current_stream()
        _hip.check(lib.per_sample_leaves(_hip.ptr(tree._nodes_dev), tree._depth, _hip.ptr(u), int(u.numel()), 0, _hip.ptr(leaves), q),
                   "per_sample_leaves")
        _hip.check(lib.sampler_map_indices(_hip.ptr(self._i2k_dev), _hip.ptr(leaves), int(u.numel()), _hip.ptr(keys), q),
                   "sampler_map_indices")
        return keys
