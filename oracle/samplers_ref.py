"""CPU oracle (test infrastructure only): the reference's sampling distributions.

Restates ``slimdqn/sample_collection/samplers.py`` of the reference:

* ``UniformRef`` (``samplers.py:13-49``): dense ``index -> key`` list plus ``key -> index`` dict;
  ``remove`` swaps the victim with the last entry and pops (``:26-37``); ``sample`` draws
  ``Generator(PCG64(seed)).integers(len, size=n)`` and maps indices to keys as int32 (``:39-49``).
* ``PrioritizedRef`` (``samplers.py:52-116``): the same maps plus a sum tree over the *local
  index*; ``add`` stores ``0 if p == 0 else p**alpha`` (``:66-73``); ``update`` does the same for a
  vector of keys (``:75-87``); ``remove`` moves the last leaf's priority into the hole and zeroes
  the last leaf in ONE two-element ``set`` (``:89-103``); ``sample`` draws
  ``uniform(0, root, n)`` and inverts the CDF with ``query`` (``:105-116``).
  The reference's ``root == 0`` branch is broken (``.keys`` on an ndarray -> AttributeError,
  ``:106-108``); the oracle reproduces the exception type.
"""
import numpy as np

from .sumtree_ref import SumTreeRef


class UniformRef:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.index_to_key = []
        self.key_to_index = {}

    def add(self, key):
        self.key_to_index[key] = len(self.index_to_key)
        self.index_to_key.append(key)

    def remove(self, key):
        assert key in self.key_to_index, ValueError(f"Key {key} not found.")
        hole = self.key_to_index.pop(key)
        last_key = self.index_to_key.pop()
        if last_key != key:  # the last entry fills the hole
            self.index_to_key[hole] = last_key
            self.key_to_index[last_key] = hole

    _index_to_key = property(lambda self: self.index_to_key)
    _key_to_index = property(lambda self: self.key_to_index)

    def draw_indices(self, size):
        assert self.index_to_key, ValueError("No keys to sample from.")
        return self.rng.integers(len(self.index_to_key), size=size)

    def sample(self, size):
        idx = self.draw_indices(size)
        return np.asarray([self.index_to_key[i] for i in idx], dtype=np.int32)


class PrioritizedRef(UniformRef):
    def __init__(self, seed, max_capacity, priority_exponent=1.0):
        super().__init__(seed)
        self.max_capacity = max_capacity
        self.alpha = priority_exponent
        self.tree = SumTreeRef(max_capacity)

    _sum_tree = property(lambda self: self.tree)

    def _shape(self, p):
        return 0.0 if p == 0.0 else p**self.alpha

    def add(self, key, priority):
        super().add(key)
        if priority is None:
            priority = 0.0
        self.tree.set(self.key_to_index[key], self._shape(priority))

    def update(self, keys, priorities):
        if not isinstance(keys, np.ndarray):
            keys = np.asarray([keys], dtype=np.int32)
        shaped = np.where(priorities == 0.0, 0.0, priorities**self.alpha)
        local = np.asarray([self.key_to_index[k] for k in keys], dtype=np.int32)
        self.tree.set(local, shaped)

    def remove(self, key):
        hole = self.key_to_index[key]
        last = len(self.index_to_key) - 1
        if hole == last:
            self.tree.set(hole, 0.0)
        else:
            self.tree.set(
                np.asarray([hole, last], dtype=np.int32),
                np.asarray([self.tree.get(last), 0.0]),
            )
        super().remove(key)

    def sample(self, size):
        if self.tree.root == 0.0:
            return super().sample(size).keys  # AttributeError, as in the reference (samplers.py:106-108)
        targets = self.rng.uniform(0.0, self.tree.root, size=size)
        local = self.tree.query(targets)
        return np.asarray([self.index_to_key[i] for i in local], dtype=np.int32)
