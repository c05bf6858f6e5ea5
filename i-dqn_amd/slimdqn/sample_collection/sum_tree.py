"""Sum tree whose node array lives in HBM (fp64), driven through the C ABI.

Drop-in for the reference's ``slimdqn/sample_collection/sum_tree.py:8-102`` (same constructor,
``set`` / ``get`` / ``root`` / ``query`` / ``max_recorded_priority``, same exception types).  The
arithmetic -- including the order of the fp64 additions, which the reference inherits from
``np.unique`` + ``np.add.at`` -- runs in ``csrc/sumtree.hip``; this file only validates arguments,
moves the (tiny) index / value vectors to the device and maps status codes to exceptions.
"""
import ctypes as C
import math

import numpy as np
import torch

from slimdqn import _hip

_MAX_SET = 4096
_MAIL_N = 4096  # targets per query launch through the host mailbox


class SumTree:
    def __init__(self, capacity: int) -> None:
        assert capacity > 0, "Capacity to sum tree must be positive."
        self._capacity = capacity
        self._depth = int(math.ceil(math.log2(capacity))) + 1
        self._first_leaf_offset = (2 ** (self._depth - 1)) - 1
        _hip.lib()  # fail loudly before touching the GPU if the extension is not built
        self._nodes_dev = torch.zeros((2**self._depth) - 1, dtype=torch.float64, device="cuda")
        self._scratch = torch.empty(_MAX_SET * 2, dtype=torch.float64, device="cuda")
        self._status = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.max_recorded_priority = 1.0
        self._mailbox = None   # host mailbox of query / sample (created on first use)
        self._pins = None      # pinned staging of `set`: two (indices, values) sets used in turn
        self._stream = None    # a stream of its own for every launch that touches the tree (use_own_stream), or None: torch's current

    def use_own_stream(self) -> None:
        """Every launch that reads or writes the tree goes to ONE high-priority stream owned by the tree instead of torch's
        current stream.  Tree operations stay ordered among themselves (that order is all their results depend on: they take
        kernel arguments and host vectors, nothing computed on another stream), but a ``query_host`` no longer queues behind
        whatever the caller has in flight -- the learner's gradient step of 0.27 ms -- before the host can read its mailbox."""
        if self._stream is None:
            self._stream = torch.cuda.Stream(priority=-1)
            self._stream.wait_stream(torch.cuda.current_stream())  # the zero-fill of the node array

    def _q(self):
        return _hip.current_stream() if self._stream is None else C.c_void_p(self._stream.cuda_stream)

    def _sync_to_current(self) -> None:
        """Before the caller's stream reads tree memory with its own operations (tests, ``sample_device``)."""
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def __del__(self):
        mb, self._mailbox = getattr(self, "_mailbox", None), None
        if mb is not None:
            try:
                _hip.lib().sampler_mailbox_destroy(mb)
            except Exception:  # interpreter shutdown
                pass

    # the reference's tests read ``_nodes`` directly (tests/test_sum_tree.py:34-37)
    @property
    def _nodes(self) -> np.ndarray:
        self._sync_to_current()
        return self._nodes_dev.cpu().numpy()

    def _launch_set(self, idx: np.ndarray, val: np.ndarray) -> None:
        # indices and values travel through pinned memory by asynchronous copies (a pageable upload blocks the host for
        # ~20 us each); a staging set is rewritten only after the event recorded behind the launch that read it
        if self._pins is None:
            self._pins = [[torch.empty(_MAX_SET, dtype=torch.int32).pin_memory(), torch.empty(_MAX_SET, dtype=torch.float64).pin_memory(),
                           torch.empty(_MAX_SET, dtype=torch.int32, device="cuda"), torch.empty(_MAX_SET, dtype=torch.float64, device="cuda"),
                           None] for _ in range(2)]
            self._pin_turn = 0
        pi, pv, i_dev, v_dev, ev = ent = self._pins[self._pin_turn]
        self._pin_turn ^= 1
        if ev is not None:
            ev.synchronize()
        n = int(idx.size)
        pi.numpy()[:n] = idx
        pv.numpy()[:n] = val
        with torch.cuda.stream(self._stream):  # (None: the current stream)
            i_dev[:n].copy_(pi[:n], non_blocking=True)
            v_dev[:n].copy_(pv[:n], non_blocking=True)
            _hip.check(
                _hip.lib().sumtree_set(_hip.ptr(self._nodes_dev), self._depth, _hip.ptr(i_dev), _hip.ptr(v_dev),
                                       int(idx.size), _hip.ptr(self._scratch), self._q()),
                "sumtree_set")
            ent[4] = ent[4] or torch.cuda.Event()
            ent[4].record()

    def set(self, indices, values) -> None:
        if isinstance(indices, (int, np.integer)) and isinstance(values, (int, float, np.floating)):
            # one leaf (what every add / remove of the prioritized sampler does): index and value travel as kernel
            # arguments, no upload -- same arithmetic as the vector path for n = 1
            assert values >= 0.0, "Values must be positive."
            if not 0 <= int(indices) < self._nodes_dev.numel() - self._first_leaf_offset:
                raise IndexError("sum tree index out of range")
            self.max_recorded_priority = max(self.max_recorded_priority, values)
            _hip.check(_hip.lib().sumtree_set_one(_hip.ptr(self._nodes_dev), self._depth, int(indices), float(values), None,
                                                  self._q()), "sumtree_set_one")
            return
        if isinstance(indices, (int, np.integer)):
            indices = np.asarray([indices], np.int32)
        if isinstance(values, (int, float, np.floating)):
            values = np.asarray([values], np.float64)
        indices, values = np.asarray(indices), np.asarray(values)
        assert indices.shape == values.shape, "Indices and values must have the same shape."
        assert (values >= 0.0).all(), "Values must be positive."
        if indices.size == 0:
            return
        if ((indices < 0) | (indices >= self._nodes_dev.numel() - self._first_leaf_offset)).any():
            raise IndexError("sum tree index out of range")
        self.max_recorded_priority = max(self.max_recorded_priority, max(values))
        idx = indices.reshape(-1).astype(np.int32)
        val = values.reshape(-1).astype(np.float64)  # f32 -> f64 is exact, as in `values - nodes[...]`
        if idx.size <= _MAX_SET:
            self._launch_set(idx, val)
            return
        # rare: more than one workgroup's worth.  De-duplicate here (first occurrence wins, ascending
        # leaves) and feed ascending chunks; every node then still accumulates in ascending-leaf order.
        uniq, first = np.unique(idx, return_index=True)
        for lo in range(0, uniq.size, _MAX_SET):
            self._launch_set(uniq[lo : lo + _MAX_SET], val[first[lo : lo + _MAX_SET]])

    def get(self, index):
        self._sync_to_current()
        if isinstance(index, (int, np.integer)):
            return float(self._nodes_dev[self._first_leaf_offset + int(index)].item())
        idx = torch.from_numpy(np.ascontiguousarray(index, dtype=np.int32).reshape(-1)).cuda()
        out = torch.empty(idx.numel(), dtype=torch.float64, device="cuda")
        _hip.check(_hip.lib().sumtree_get(_hip.ptr(self._nodes_dev), self._depth, _hip.ptr(idx), idx.numel(),
                                          _hip.ptr(out), _hip.current_stream()), "sumtree_get")
        return out.cpu().numpy().reshape(np.shape(index))

    @property
    def root(self) -> float:
        self._sync_to_current()
        return float(self._nodes_dev[0].item())

    def query_device(self, targets_dev: torch.Tensor) -> torch.Tensor:
        """Device-to-device query (no host sync): int32 leaf indices; status bits land in ``_status``."""
        self._sync_to_current()
        out = torch.empty(targets_dev.numel(), dtype=torch.int32, device="cuda")
        _hip.check(_hip.lib().sumtree_query(_hip.ptr(self._nodes_dev), self._depth, _hip.ptr(targets_dev),
                                            targets_dev.numel(), _hip.ptr(out), _hip.ptr(self._status),
                                            _hip.current_stream()), "sumtree_query")
        return out

    def query_host(self, values: np.ndarray, scale_by_root: bool = False, index_to_key=None, n_live: int = -1):
        """ONE launch + ONE host read (a polled mailbox in mapped host memory): the leaves of `values` (targets, or --
        scale_by_root -- uniforms in [0, 1) turned into numpy's ``uniform(0, root)`` on the device), the keys
        ``index_to_key[leaf]`` when a device map is given (``n_live`` = its valid entries: a leaf behind them sets status
        bit 2 and gets key -1), the root and the status bits."""
        vals = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        lib = _hip.lib()
        if self._mailbox is None:
            mb = C.c_void_p()
            _hip.check(lib.sampler_mailbox_create(_MAIL_N, C.byref(mb)), "sampler_mailbox_create")
            self._mailbox = mb
        leaves = np.empty(vals.size, np.int32)
        keys = np.empty(vals.size, np.int32) if index_to_key is not None else None
        root, status, st_all = C.c_double(), C.c_int32(), 0
        for lo in range(0, vals.size, _MAIL_N):
            n = min(_MAIL_N, vals.size - lo)
            _hip.check(lib.sumtree_query_host(
                _hip.ptr(self._nodes_dev), self._depth, C.c_void_p(vals[lo:].ctypes.data), n, 1 if scale_by_root else 0,
                _hip.ptr(index_to_key), int(n_live), self._mailbox, C.c_void_p(leaves[lo:].ctypes.data),
                C.c_void_p(keys[lo:].ctypes.data) if keys is not None else None, C.byref(root), C.byref(status),
                self._q()), "sumtree_query_host")
            st_all |= status.value
        return leaves, keys, root.value, st_all

    def query(self, targets):
        if isinstance(targets, (int, float)):
            targets = np.asarray([targets], np.float64)
        targets = np.asarray(targets)
        if targets.size == 0:
            return np.empty(targets.shape, np.int32)
        leaves, _, root, status = self.query_host(targets)
        if status & 1:
            raise ValueError(f"Targets must be in the interval [0.0, {root}).")
        assert not (status & 2), "sum tree traversal: target not below its node (sum_tree.py:81)"
        return leaves.reshape(targets.shape)
