"""CPU oracle (test infrastructure only): the reference's vectorised fp64 sum tree.

Restates ``slimdqn/sample_collection/sum_tree.py:8-102`` of the reference:

* geometry (``sum_tree.py:11-18``): ``depth = ceil(log2(capacity)) + 1``, leaves start at
  ``2**(depth-1) - 1``, ``2**depth - 1`` float64 nodes, ``max_recorded_priority`` starts at 1.
* ``set`` (``sum_tree.py:20-47``): deltas are taken against the *current* leaf values before
  de-duplication; duplicates keep their FIRST occurrence (``np.unique(return_index=True)``) and
  the survivors are processed in ascending leaf order; each tree level is one ``np.add.at`` --
  an unbuffered, strictly sequential accumulation, so nodes shared by several updated leaves
  receive ``((node + d_0) + d_1) + ...`` in ascending-leaf order.  That order is what makes
  the result bit-reproducible and is what the HIP kernel must match.
* ``query`` (``sum_tree.py:58-102``): ``ValueError`` unless ``0 <= t < root``; level-synchronous
  descent, going left when ``t < left`` and otherwise subtracting ``left`` and going right.

The loops here are deliberately scalar (a second, structurally different statement of the same
arithmetic) so that agreement with reference traces is a meaningful check.
"""
import math

import numpy as np


class SumTreeRef:
    def __init__(self, capacity):
        assert capacity > 0, "Capacity to sum tree must be positive."
        self.capacity = int(capacity)
        self.depth = int(math.ceil(math.log2(capacity))) + 1
        self.first_leaf = (1 << (self.depth - 1)) - 1
        self.nodes = np.zeros((1 << self.depth) - 1, dtype=np.float64)
        self.max_recorded_priority = 1.0

    # -- helpers ---------------------------------------------------------------------------
    @staticmethod
    def _as_arrays(indices, values):
        if isinstance(indices, (int, np.integer)):
            indices = np.asarray([indices], np.int32)
        if isinstance(values, (int, float, np.floating)):
            values = np.asarray([values], np.float64)
        indices = np.asarray(indices)
        values = np.asarray(values)
        return indices, values

    # -- API -------------------------------------------------------------------------------
    def set(self, indices, values):
        indices, values = self._as_arrays(indices, values)
        assert indices.shape == values.shape, "Indices and values must have the same shape."
        assert (values >= 0.0).all(), "Values must be positive."
        self.max_recorded_priority = max(self.max_recorded_priority, max(values))
        # delta against the present leaf value; float32 inputs are promoted exactly like
        # numpy does in `values - nodes[...]` (f32 - f64 -> f64).
        leaves = self.first_leaf + indices.astype(np.int64)
        deltas = values.astype(np.float64) - self.nodes[leaves]
        # first occurrence wins, ascending leaf order
        first_seen = {}
        for pos, leaf in enumerate(leaves.tolist()):
            if leaf not in first_seen:
                first_seen[leaf] = pos
        order = sorted(first_seen)
        cur = [int(x) for x in order]
        dl = [float(deltas[first_seen[leaf]]) for leaf in order]
        nodes = self.nodes
        for _ in range(self.depth):
            for node, d in zip(cur, dl):  # strictly sequential -> ((n + d0) + d1) + ...
                nodes[node] = nodes[node] + d
            if cur and cur[0] == 0 and all(c == 0 for c in cur):
                break
            cur = [(c - 1) // 2 for c in cur]

    def get(self, index):
        return self.nodes[self.first_leaf + index]

    # reference attribute names (tests reach into them: tests/test_sum_tree.py:34-37,80-81)
    _nodes = property(lambda self: self.nodes)
    _depth = property(lambda self: self.depth)
    _first_leaf_offset = property(lambda self: self.first_leaf)

    @property
    def root(self):
        return self.nodes[0]

    def query(self, targets):
        scalar = isinstance(targets, (int, float))
        t = np.asarray([targets], np.float64) if scalar else np.asarray(targets)
        if not ((t >= 0) & (t < self.root)).all():
            raise ValueError(f"Targets must be in the interval [0.0, {self.root}).")
        out = np.zeros(t.shape, dtype=np.int32)
        for n, target in enumerate(t.astype(np.float64).tolist()):
            node = 0
            while node < self.first_leaf:
                # the reference asserts this invariant at every level (sum_tree.py:81); rounding
                # drift in the inner nodes can break it for targets within an ulp-scale of a boundary
                assert target < float(self.nodes[node])
                left = 2 * node + 1
                left_sum = float(self.nodes[left])
                if target < left_sum:
                    node = left
                else:
                    target = target - left_sum
                    node = left + 1
            out[n] = node - self.first_leaf
        return out
