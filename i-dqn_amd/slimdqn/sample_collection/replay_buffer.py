"""Replay buffer whose elements live uncompressed in HBM.

Drop-in for the reference's ``slimdqn/sample_collection/replay_buffer.py`` (same constructor and
``add`` / ``sample`` / ``update``, ``add_count``, ``_memory``, ``_sampling_distribution``,
``_clipping``).  Differences are storage only:

* the reference keeps snappy-compressed elements in a host ``OrderedDict`` keyed by ``add_count``
  and evicts FIFO (``:202-213``); here element ``key`` occupies slot ``key % max_capacity`` of a
  device array ``[capacity][2][obs_bytes]`` (state, next_state) -- the same FIFO, no dictionary;
  ``compress`` is accepted and ignored (56 KB/slot x 1 M slots = 56 GB fits 288 GB of HBM3E);
* ``sample()`` (``:215-230``) gathers on the device (``csrc/replay.hip``) and returns a
  ``ReplayElement`` of device arrays; ``np.asarray(field)`` copies back when a test wants to look.

The n-step / frame-stack accumulator (``:103-200``) is host logic here as in the reference
(``TrajectoryAccumulator``, pure Python, no GPU needed).
"""
import collections
import typing
from typing import Any, Optional

import numpy as np

from slimdqn.sample_collection import ReplayItemID


class TransitionElement(typing.NamedTuple):
    observation: Optional[np.ndarray]
    action: int
    reward: float
    is_terminal: bool
    episode_end: bool = False


class ReplayElement(typing.NamedTuple):
    """(state, action, reward, next_state, is_terminal, episode_end) -- reference ``:26-34``."""

    state: Any
    action: Any
    reward: Any
    next_state: Any
    is_terminal: Any
    episode_end: Any


class DevArray:
    """A device tensor that numpy can look at (``np.asarray`` copies to the host)."""

    def __init__(self, tensor):
        self.tensor = tensor

    shape = property(lambda self: tuple(self.tensor.shape))

    def __array__(self, dtype=None, copy=None):
        out = self.tensor.cpu().numpy()
        return out.astype(dtype) if dtype is not None else out

    def __getitem__(self, item):
        return self.tensor[item].cpu().numpy()[()]

    def __len__(self):
        return self.tensor.shape[0]


class TrajectoryAccumulator:
    """n-step + frame-stack window -> replay elements (reference ``:103-200``); host only."""

    def __init__(self, stack_size: int, update_horizon: int, gamma: float) -> None:
        self.stack_size, self.n, self.gamma = stack_size, update_horizon, gamma
        self.window = collections.deque(maxlen=update_horizon + stack_size)

    def _emit(self):
        w = self.window
        size, tail = len(w), w[-1]
        if not (size > self.n or (size > 1 and tail.is_terminal)):
            return None
        # a terminal that arrives early shortens the horizon (":114-117")
        horizon = size - 1 if (tail.is_terminal and size <= self.n) else self.n
        frame = np.asarray(tail.observation)
        state = np.zeros(frame.shape + (self.stack_size,), frame.dtype)
        nxt = np.zeros_like(state)
        last_s = size - horizon - 1  # window position of the newest frame of `state`
        last_n = size - 1
        reward = 0.0
        for t, tr in enumerate(w):
            if last_s <= t <= last_s + self.n - 1:
                reward += tr.reward * (self.gamma ** (t - last_s))
            ch = t - (last_s - self.stack_size + 1)
            if 0 <= ch < self.stack_size:
                state[..., ch] = tr.observation
            ch = t - (last_n - self.stack_size + 1)
            if 0 <= ch < self.stack_size:
                nxt[..., ch] = tr.observation
        done = w[last_n].is_terminal
        return ReplayElement(state, w[last_s].action, reward, nxt, done, done)  # episode_end := is_terminal (":146")

    def push(self, transition: TransitionElement):
        self.window.append(transition)
        if transition.is_terminal:
            while (el := self._emit()) is not None:
                yield el
                self.window.popleft()
            self.window.clear()
            return
        el = self._emit()
        if el is not None:
            yield el
        if transition.episode_end:
            self.window.clear()


class _MemoryView:
    """Read-only mapping ``key -> ReplayElement`` over the device store (what tests call ``_memory``)."""

    def __init__(self, rb):
        self._rb = rb

    def keys(self):
        lo = max(0, self._rb.add_count - self._rb._max_capacity)
        return range(lo, self._rb.add_count)

    __iter__ = lambda self: iter(self.keys())
    __len__ = lambda self: len(self.keys())

    def __contains__(self, key):
        return key in self.keys()

    def __getitem__(self, key):
        if key not in self.keys():
            raise KeyError(key)
        rb = self._rb
        slot = key % rb._max_capacity
        frames = rb._store[slot].cpu().numpy()
        s = np.frombuffer(frames[0].tobytes(), dtype=rb._obs_dtype).reshape(rb._obs_shape)
        n = np.frombuffer(frames[1].tobytes(), dtype=rb._obs_dtype).reshape(rb._obs_shape)
        return ReplayElement(s, int(rb._action[slot]), float(rb._reward64[slot]), n,
                             bool(rb._terminal[slot]), bool(rb._terminal[slot]))


class ReplayBuffer:
    def __init__(self, sampling_distribution, batch_size: int, max_capacity: int, stack_size: int = 4,
                 update_horizon: int = 1, gamma: float = 0.99, checkpoint_duration: int = 4, compress: bool = True,
                 clipping: callable = None):
        self.add_count = 0
        self._max_capacity = max_capacity
        self._compress = compress  # accepted for signature parity; the HBM store is uncompressed
        self._sampling_distribution = sampling_distribution
        self._checkpoint_duration = checkpoint_duration
        self._batch_size = batch_size
        self._stack_size, self._update_horizon, self._gamma = stack_size, update_horizon, gamma
        self._clipping = clipping
        self._accumulator = TrajectoryAccumulator(stack_size, update_horizon, gamma)
        self._store = None  # device [capacity][2][obs_bytes] uint8, allocated at the first add
        self._memory = _MemoryView(self)
        self._stage = {}

    # ---- storage -------------------------------------------------------------------------------
    def _allocate(self, state: np.ndarray) -> None:
        import torch

        from slimdqn import _hip

        _hip.lib()  # no extension -> no replay buffer (there is no host store to fall back to)
        self._obs_shape, self._obs_dtype = state.shape, state.dtype
        self._obs_bytes = int(state.nbytes)
        cap = self._max_capacity
        self._store = torch.empty((cap, 2, self._obs_bytes), dtype=torch.uint8, device="cuda")
        self._action_dev = torch.zeros(cap, dtype=torch.int32, device="cuda")
        self._reward_dev = torch.zeros(cap, dtype=torch.float32, device="cuda")
        self._terminal_dev = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        # host mirrors of the scalars (cheap; `_memory[key]` and logging read them)
        self._action = np.zeros(cap, np.int32)
        self._reward64 = np.zeros(cap, np.float64)
        self._terminal = np.zeros(cap, np.uint8)
        self._pin = torch.empty((2, self._obs_bytes), dtype=torch.uint8).pin_memory()

    def _write(self, key: int, el: ReplayElement) -> None:
        import torch

        if self._store is None:
            self._allocate(el.state)
        slot = key % self._max_capacity
        torch.cuda.current_stream().synchronize()  # the pinned staging buffer may still be in flight
        self._pin[0].copy_(torch.from_numpy(np.ascontiguousarray(el.state).view(np.uint8).reshape(-1)))
        self._pin[1].copy_(torch.from_numpy(np.ascontiguousarray(el.next_state).view(np.uint8).reshape(-1)))
        self._store[slot].copy_(self._pin, non_blocking=True)
        self._action[slot], self._reward64[slot], self._terminal[slot] = el.action, el.reward, el.is_terminal
        self._action_dev[slot] = int(el.action)
        self._reward_dev[slot] = float(el.reward)  # f64 -> f32, the downcast jit applies to batch.reward
        self._terminal_dev[slot] = int(bool(el.is_terminal))

    # ---- reference API -----------------------------------------------------------------------------
    def add(self, transition: TransitionElement, **kwargs: Any) -> None:
        for el in self._accumulator.push(transition):
            key = ReplayItemID(self.add_count)
            self._write(key, el)
            self._sampling_distribution.add(key, **kwargs)
            self.add_count += 1
            if self.add_count > self._max_capacity:  # FIFO: the oldest key leaves (":211-213")
                self._sampling_distribution.remove(self.add_count - 1 - self._max_capacity)

    def _staging(self, size: int):
        import torch

        if size not in self._stage:
            self._stage[size] = dict(
                slots=torch.empty(size, dtype=torch.int32, device="cuda"),
                state=torch.empty((size, self._obs_bytes), dtype=torch.uint8, device="cuda"),
                next_state=torch.empty((size, self._obs_bytes), dtype=torch.uint8, device="cuda"),
                action=torch.empty(size, dtype=torch.int32, device="cuda"),
                reward=torch.empty(size, dtype=torch.float32, device="cuda"),
                terminal=torch.empty(size, dtype=torch.uint8, device="cuda"),
            )
        return self._stage[size]

    def sample(self, size=None) -> ReplayElement:
        import torch

        from slimdqn import _hip

        assert self.add_count, ValueError("No samples in replay buffer!")
        if size is None:
            size = self._batch_size
        keys = self._sampling_distribution.sample(size)
        st = self._staging(size)
        st["slots"].copy_(torch.from_numpy((np.asarray(keys, np.int64) % self._max_capacity).astype(np.int32)))
        lib, q = _hip.lib(), _hip.current_stream()
        _hip.check(lib.replay_gather(_hip.ptr(self._store), self._obs_bytes, _hip.ptr(st["slots"]), size,
                                     _hip.ptr(st["state"]), _hip.ptr(st["next_state"]), q), "replay_gather")
        _hip.check(lib.replay_gather_scalars(_hip.ptr(self._action_dev), _hip.ptr(self._reward_dev),
                                             _hip.ptr(self._terminal_dev), _hip.ptr(st["slots"]), size,
                                             _hip.ptr(st["action"]), _hip.ptr(st["reward"]), _hip.ptr(st["terminal"]), q),
                   "replay_gather_scalars")
        tdt = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32,
               np.dtype(np.float64): torch.float64, np.dtype(np.int64): torch.int64,
               np.dtype(np.int32): torch.int32}[np.dtype(self._obs_dtype)]
        shape = (size,) + tuple(self._obs_shape)
        return ReplayElement(
            state=DevArray(st["state"].view(tdt).view(shape)),
            action=DevArray(st["action"]),
            reward=DevArray(st["reward"]),
            next_state=DevArray(st["next_state"].view(tdt).view(shape)),
            is_terminal=DevArray(st["terminal"]),
            episode_end=DevArray(st["terminal"]),
        )

    def update(self, keys, **kwargs: Any) -> None:
        self._sampling_distribution.update(keys, **kwargs)
