"""Replay buffer whose frames live in HBM, written once each.

Drop-in for the reference's ``slimdqn/sample_collection/replay_buffer.py`` (same constructor and
``add`` / ``sample`` / ``update``, ``add_count``, ``_memory``, ``_sampling_distribution``,
``_clipping``).  What differs is storage:

* the reference materialises every element's two frame stacks on the host, snappy-compresses them and keeps them in
  an ``OrderedDict`` keyed by ``add_count`` with FIFO eviction (``:119-137,202-213``).  Here each environment frame
  goes to HBM ONCE -- ring slot ``transition index % n_frames`` of ``[n_frames][frame_bytes]`` -- and an element is
  a row of 8 int32 (which frames its stacks end at, how many are valid, action, reward, terminal) at slot
  ``key % max_capacity``: the same FIFO, 1/8 of the bytes (7 KB instead of 56 KB per Atari element: a 1 M buffer is
  7 GB of the 288 GB), and one 7 KB host-to-device copy per environment step.  ``compress`` is accepted and ignored;
* ``sample()`` (``:215-230``) assembles the stacks on the device (``replay_gather_stacked`` in ``csrc/replay.hip``:
  gather + zero padding before the episode start + ``np.stack`` in one launch) and returns a ``ReplayElement`` of
  device arrays; ``np.asarray(field)`` copies back when a test wants to look.

The n-step / frame-stack window logic (``:103-200``) is host integer logic (``TrajectoryAccumulator``): it decides
WHICH frames and rewards make an element; it never touches pixel data in the buffer's path.
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


class ElementPlan(typing.NamedTuple):
    """An element as window positions: its stacks end at ``last_s`` / ``last_n`` (older positions fill the earlier
    channels, positions < 0 are zero frames), ``reward`` is the discounted n-step sum."""

    last_s: int
    last_n: int
    action: Any
    reward: float
    done: bool


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
    """n-step + frame-stack window -> replay elements (reference ``:103-200``); host only.

    ``plan`` yields ``(ElementPlan, window size)`` -- positions, not pixels; ``push`` materialises the stacks from the
    observations held in the window (used by host-side tests and anyone who wants the reference's element)."""

    def __init__(self, stack_size: int, update_horizon: int, gamma: float) -> None:
        self.stack_size, self.n, self.gamma = stack_size, update_horizon, gamma
        self.window = collections.deque(maxlen=update_horizon + stack_size)

    def _plan(self):
        w = self.window
        size, tail = len(w), w[-1]
        if not (size > self.n or (size > 1 and tail.is_terminal)):
            return None
        # a terminal that arrives early shortens the horizon (":114-117")
        horizon = size - 1 if (tail.is_terminal and size <= self.n) else self.n
        last_s = size - horizon - 1  # window position of the newest frame of `state`
        last_n = size - 1
        reward = 0.0
        for t, tr in enumerate(w):
            if last_s <= t <= last_s + self.n - 1:
                reward += tr.reward * (self.gamma ** (t - last_s))
        return ElementPlan(last_s, last_n, w[last_s].action, reward, w[last_n].is_terminal)

    def _materialise(self, plan: ElementPlan) -> ReplayElement:
        w = self.window
        frame = np.asarray(w[-1].observation)
        state = np.zeros(frame.shape + (self.stack_size,), frame.dtype)
        nxt = np.zeros_like(state)
        for t, tr in enumerate(w):
            ch = t - (plan.last_s - self.stack_size + 1)
            if 0 <= ch < self.stack_size:
                state[..., ch] = tr.observation
            ch = t - (plan.last_n - self.stack_size + 1)
            if 0 <= ch < self.stack_size:
                nxt[..., ch] = tr.observation
        # episode_end := is_terminal (":146")
        return ReplayElement(state, plan.action, plan.reward, nxt, plan.done, plan.done)

    def _steps(self, transition: TransitionElement, emit):
        self.window.append(transition)
        if transition.is_terminal:
            while (plan := self._plan()) is not None:
                yield emit(plan)
                self.window.popleft()
            self.window.clear()
            return
        plan = self._plan()
        if plan is not None:
            yield emit(plan)
        if transition.episode_end:
            self.window.clear()

    def push(self, transition: TransitionElement):
        """Yields the reference's ReplayElements (materialised stacks)."""
        return self._steps(transition, self._materialise)

    def plan(self, transition: TransitionElement):
        """Yields ``(ElementPlan, window size when it was made)``; the window's newest entry is the pushed transition."""
        return self._steps(transition, lambda p: (p, len(self.window)))


class _MemoryView:
    """Read-only mapping ``key -> ReplayElement`` over the device store (what tests call ``_memory``)."""

    def __init__(self, rb):
        import weakref

        self._rb = weakref.proxy(rb)  # no reference cycle: dropping the buffer frees its HBM at once

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
        rb = self._rb  # (`keys()` above read add_count: a postponed add has been applied)
        slot = key % rb._max_capacity
        el = rb._gather(np.asarray([slot], np.int32))
        return ReplayElement(np.asarray(el.state)[0], int(rb._action[slot]), float(rb._reward64[slot]),
                             np.asarray(el.next_state)[0], bool(rb._meta[slot, 6]), bool(rb._meta[slot, 6]))


class ReplayBuffer:
    STAGING = 64  # pinned host frames in flight towards the ring

    # `add_count` (and through it `_memory`) is read by callers at any time: a read first applies a postponed add
    # (add_deferred), so that every observer sees the buffer `add` would have left -- mid-epoch callbacks included.
    @property
    def add_count(self) -> int:
        if getattr(self, "_deferred", None) is not None:
            self.flush_deferred()
        return self._add_count

    @add_count.setter
    def add_count(self, value: int) -> None:
        self._add_count = value

    def __init__(self, sampling_distribution, batch_size: int, max_capacity: int, stack_size: int = 4,
                 update_horizon: int = 1, gamma: float = 0.99, checkpoint_duration: int = 4, compress: bool = True,
                 clipping: callable = None):
        self._add_count = 0
        self._max_capacity = max_capacity
        self._compress = compress  # accepted for signature parity; the HBM store is uncompressed
        self._sampling_distribution = sampling_distribution
        self._checkpoint_duration = checkpoint_duration
        self._batch_size = batch_size
        self._stack_size, self._update_horizon, self._gamma = stack_size, update_horizon, gamma
        self._clipping = clipping
        self._accumulator = TrajectoryAccumulator(stack_size, update_horizon, gamma)
        self._frames = None  # device [n_frames][frame_bytes] uint8, allocated at the first add
        self._memory = _MemoryView(self)
        self._stage = {}
        # sample() hands out FRESH device arrays by default, like the reference (replay_buffer.py:229 stacks new arrays).
        # The trainer loop, which consumes a batch before it draws the next one, sets this to True: sample() then returns
        # the same ReplayElement object per staging set (views that the second-next sample of that size overwrites).
        self.reuse_sample_buffers = False
        self._t = 0  # transitions pushed so far == index of the next frame
        self._flushed = 0  # elements whose metadata row is already on the device

    # ---- storage -------------------------------------------------------------------------------
    def _allocate(self, frame: np.ndarray) -> None:
        import torch

        from slimdqn import _hip

        _hip.lib()  # no extension -> no replay buffer (there is no host store to fall back to)
        self._frame_shape, self._obs_dtype = tuple(frame.shape), frame.dtype
        self._frame_elems, self._itemsize = int(frame.size), int(frame.dtype.itemsize)
        self._frame_bytes = self._frame_elems * self._itemsize
        self._obs_shape = self._frame_shape + (self._stack_size,)
        cap = self._max_capacity
        # Every alive element's frames must still be in the ring.  An element is made per transition except for the
        # few transitions before a truncated (non-terminal) episode end, so capacity + window + slack frames cover
        # capacity elements; `_upload_frame` grows the ring in the rare case the slack runs out.
        self._n_frames = cap + self._update_horizon + self._stack_size + max(64, cap // 16)
        self._frames = torch.empty((self._n_frames, self._frame_bytes), dtype=torch.uint8, device="cuda")
        # host copy of the element rows (see replay_gather_stacked), in PINNED memory: new rows go to the device by
        # asynchronous copies.  A row is rewritten only `capacity` adds after it was written; `_write` waits for the last
        # upload first in the (tiny-buffer) case that it could still be in flight.
        self._meta_pin = torch.zeros((cap, 8), dtype=torch.int32).pin_memory()
        self._meta = self._meta_pin.numpy()
        self._meta_dev = torch.zeros((cap, 8), dtype=torch.int32, device="cuda")
        self._meta_ev, self._meta_ev_lo = None, 0  # event behind the last upload, oldest key it may still be reading
        self._meta_ev_obj = None
        self._first_frame = np.zeros(cap, np.int64)  # oldest frame (transition index) an element refers to
        # host mirrors of the scalars (cheap; `_memory[key]` and logging read them)
        self._action = np.zeros(cap, np.int64)
        self._reward64 = np.zeros(cap, np.float64)
        self._pin = torch.empty((self.STAGING, self._frame_bytes), dtype=torch.uint8).pin_memory()
        self._pin_np = self._pin.numpy()
        self._pin_events = [None, None]  # one per half of the staging ring
        self._pin_ptr = self._pin.data_ptr()

    def _grow_ring(self, oldest_needed: int) -> None:
        """Doubles the frame ring, keeping frames [oldest_needed, _t) (transition indices) in place modulo the new size."""
        import torch

        from slimdqn import _hip

        if self._meta_ev is not None:  # the rows are rewritten below: no upload may still be reading them
            self._meta_ev.synchronize()
            self._meta_ev = None
        old_n, new_n = self._n_frames, 2 * self._n_frames
        new = torch.empty((new_n, self._frame_bytes), dtype=torch.uint8, device="cuda")
        _hip.check(_hip.lib().replay_ring_regrow(_hip.ptr(self._frames), old_n, _hip.ptr(new), new_n, oldest_needed,
                                                 self._t - oldest_needed, self._frame_bytes, _hip.current_stream()),
                   "replay_ring_regrow")
        # frame slots are stored in the element rows: re-derive them for the alive elements (all at once)
        keys = np.arange(max(0, self.add_count - self._max_capacity), self.add_count, dtype=np.int64)
        if keys.size:
            slots = keys % self._max_capacity
            newest_s = self._first_frame[slots] + self._meta[slots, 1] - 1
            horizon = (self._meta[slots, 2].astype(np.int64) - self._meta[slots, 0]) % old_n
            self._meta[slots, 0] = newest_s % new_n
            self._meta[slots, 2] = (newest_s + horizon) % new_n
        self._frames, self._n_frames = new, new_n
        self._flushed = max(0, self.add_count - self._max_capacity)  # every alive row goes to the device again

    def _upload_frame(self, observation) -> int:
        """The newest frame goes to ring slot ``t % n_frames`` (async copy from a pinned staging slot); returns t."""
        import torch

        frame = np.ascontiguousarray(observation)
        if self._frames is None:
            self._allocate(frame)
        assert frame.shape == self._frame_shape and frame.dtype == self._obs_dtype, "observation shape / dtype changed"
        t = self._t
        if self.add_count:
            oldest_key = max(0, self.add_count - self._max_capacity)
            oldest_needed = int(self._first_frame[oldest_key % self._max_capacity])
            if t - oldest_needed >= self._n_frames:  # the slot still holds a frame an alive element needs
                self._grow_ring(oldest_needed)
        # one C call (hipMemcpyAsync) per frame; the staging ring is guarded by ONE event per half: before the first slot
        # of a half is rewritten, the copies of that half's previous round (issued >= STAGING / 2 frames ago) have passed
        from slimdqn import _hip

        i, half = t % self.STAGING, self.STAGING // 2
        if i % half == 0:
            h = i // half
            if self._pin_events[h] is not None:
                self._pin_events[h].synchronize()
        self._pin_np[i] = frame.view(np.uint8).reshape(-1)
        _hip.check(_hip.lib().replay_add_frame(_hip.ptr(self._frames), t % self._n_frames, self._frame_bytes,
                                               self._pin_ptr + i * self._frame_bytes, _hip.current_stream()), "replay_add_frame")
        if i % half == half - 1:  # last slot of a half: mark the point the stream has to pass before the half is reused
            ev = torch.cuda.Event()
            ev.record()
            self._pin_events[i // half] = ev
        self._t = t + 1
        return t

    def _write(self, key: int, plan: ElementPlan, window_size: int, t_now: int) -> None:
        slot = key % self._max_capacity
        if self._meta_ev is not None and key >= self._meta_ev_lo + self._max_capacity:  # (buffers of a few elements only)
            self._meta_ev.synchronize()
            self._meta_ev = None
        to_t = lambda pos: t_now - (window_size - 1 - pos)  # window position -> transition index
        valid_s, valid_n = min(self._stack_size, plan.last_s + 1), min(self._stack_size, plan.last_n + 1)
        row = self._meta[slot]
        row[0], row[1] = to_t(plan.last_s) % self._n_frames, valid_s
        row[2], row[3] = to_t(plan.last_n) % self._n_frames, valid_n
        row[4] = int(plan.action)
        row[5] = np.float32(plan.reward).view(np.int32)  # f64 -> f32, the downcast jit applies to batch.reward
        row[6] = int(bool(plan.done))
        self._first_frame[slot] = to_t(plan.last_s) - valid_s + 1
        self._action[slot], self._reward64[slot] = plan.action, plan.reward

    def _flush_meta(self) -> None:
        """Element rows written since the last sample go to the device: FIFO slots are contiguous modulo capacity."""
        import torch

        from slimdqn import _hip

        lo, hi, cap = self._flushed, self.add_count, self._max_capacity
        if hi == lo:
            return
        lib, q, src, dst = _hip.lib(), _hip.current_stream(), self._meta_pin.data_ptr(), self._meta_dev.data_ptr()

        def rows(a, b):  # one asynchronous copy from the pinned mirror (the byte-range form of replay_add_frame)
            _hip.check(lib.replay_add_frame(dst + 32 * a, 0, 32 * (b - a), src + 32 * a, q), "replay_add_frame")

        if hi - lo >= cap:
            rows(0, cap)
        else:
            a, b = lo % cap, hi % cap
            if a < b:
                rows(a, b)
            else:
                rows(a, cap)
                if b:
                    rows(0, b)
        if self._meta_ev is None:
            self._meta_ev_lo = lo
            if self._meta_ev_obj is None:
                self._meta_ev_obj = torch.cuda.Event()
            self._meta_ev = self._meta_ev_obj
        self._meta_ev.record()  # (re-recorded: the event now stands behind every upload issued so far)
        self._flushed = hi

    # ---- reference API -----------------------------------------------------------------------------
    def add_deferred(self, transition: TransitionElement, **kwargs: Any) -> None:
        """``add`` whose work is postponed to the next ``flush_deferred`` (the trainer calls it while the GPU computes the
        next action); ``add`` / ``sample`` / ``update`` flush first, so every observable state is the one ``add`` gives.
        The frame is copied now: the environment may reuse its buffer on the next step or reset."""
        self.flush_deferred()
        self._deferred = (transition._replace(observation=np.array(transition.observation, copy=True)), kwargs)

    def flush_deferred(self) -> None:
        d = getattr(self, "_deferred", None)
        if d is not None:
            self._deferred = None
            self.add(d[0], **d[1])

    def add(self, transition: TransitionElement, **kwargs: Any) -> None:
        if getattr(self, "_deferred", None) is not None:
            self.flush_deferred()
        t_now = self._upload_frame(transition.observation)
        light = transition._replace(observation=None)  # the window decides positions; pixels stay on the device
        for plan, window_size in self._accumulator.plan(light):
            key = ReplayItemID(self.add_count)
            self._write(key, plan, window_size, t_now)
            self._sampling_distribution.add(key, **kwargs)
            self.add_count += 1
            if self.add_count > self._max_capacity:  # FIFO: the oldest key leaves (":211-213")
                self._sampling_distribution.remove(self.add_count - 1 - self._max_capacity)

    def _staging(self, size: int):
        import torch

        # Two staging sets per batch size, used in turn: the ReplayElement a call returns holds views of one set and stays
        # valid until the SECOND-next sample of the same size (the reference returns fresh arrays every time; a caller that
        # keeps more than two batches alive has to copy them).
        if size not in self._stage:
            tdt = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32,
                   np.dtype(np.float64): torch.float64, np.dtype(np.int64): torch.int64,
                   np.dtype(np.int32): torch.int32}[np.dtype(self._obs_dtype)]
            shape = (size,) + tuple(self._obs_shape)
            sets = []
            for _ in range(2):
                st = dict(
                    slots=torch.empty(size, dtype=torch.int32, device="cuda"),
                    slots_pin=torch.empty(size, dtype=torch.int32).pin_memory(), slots_ev=None,
                    state=torch.empty((size, self._frame_bytes * self._stack_size), dtype=torch.uint8, device="cuda"),
                    next_state=torch.empty((size, self._frame_bytes * self._stack_size), dtype=torch.uint8, device="cuda"),
                    action=torch.empty(size, dtype=torch.int32, device="cuda"),
                    reward=torch.empty(size, dtype=torch.float32, device="cuda"),
                    terminal=torch.empty(size, dtype=torch.uint8, device="cuda"),
                )
                st["slots_np"] = st["slots_pin"].numpy()
                # the element handed out and the output pointers of the gather are the same every time the set is used
                st["element"] = ReplayElement(
                    state=DevArray(st["state"].view(tdt).view(shape)), action=DevArray(st["action"]),
                    reward=DevArray(st["reward"]), next_state=DevArray(st["next_state"].view(tdt).view(shape)),
                    is_terminal=DevArray(st["terminal"]), episode_end=DevArray(st["terminal"]))
                sets.append(st)
            self._stage[size] = sets + [0]
        sets = self._stage[size]
        sets[2] ^= 1
        return sets[sets[2]]

    def _gather(self, slots: np.ndarray) -> ReplayElement:
        import torch

        st = self._staging(int(slots.size))
        from slimdqn import _hip

        if st["slots_ev"] is None:
            st["slots_ev"] = torch.cuda.Event()
        else:  # the set's previous upload (two samples ago) has long passed; make sure
            st["slots_ev"].synchronize()
        st["slots_np"][:] = slots
        _hip.check(_hip.lib().replay_add_frame(st["slots"].data_ptr(), 0, 4 * int(slots.size), st["slots_pin"].data_ptr(),
                                               _hip.current_stream()), "replay_add_frame")
        st["slots_ev"].record()
        return self._gather_device(st["slots"], st)

    def _gather_device(self, slots_dev, st=None) -> ReplayElement:
        """Stacked gather for slots that are already on the device (int32 tensor): no host round trip."""
        from slimdqn import _hip

        size = int(slots_dev.numel())
        self._flush_meta()
        st = st if st is not None else self._staging(size)
        if slots_dev.data_ptr() != st["slots"].data_ptr():
            st["slots"].copy_(slots_dev)
        if "gargs" not in st:
            st["gargs"] = (_hip.ptr(self._meta_dev), _hip.ptr(st["slots"]), size, _hip.ptr(st["state"]), _hip.ptr(st["next_state"]),
                           _hip.ptr(st["action"]), _hip.ptr(st["reward"]), _hip.ptr(st["terminal"]))
        _hip.check(_hip.lib().replay_gather_stacked(
            _hip.ptr(self._frames), self._n_frames, self._frame_elems, self._itemsize, self._stack_size, *st["gargs"],
            _hip.current_stream()), "replay_gather_stacked")
        if self.reuse_sample_buffers:
            return st["element"]
        e = st["element"]  # detached copies (stream-ordered device copies behind the gather)
        return ReplayElement(state=DevArray(e.state.tensor.clone()), action=DevArray(e.action.tensor.clone()),
                             reward=DevArray(e.reward.tensor.clone()), next_state=DevArray(e.next_state.tensor.clone()),
                             is_terminal=DevArray(e.is_terminal.tensor.clone()), episode_end=DevArray(e.episode_end.tensor.clone()))

    def sample(self, size=None) -> ReplayElement:
        return self._gather(self.sample_slots(size))

    def sample_slots(self, size=None) -> np.ndarray:
        """The first half of ``sample()`` (``:215-222``): the sampler's draw, as element slots (int32, host).  ``_gather(slots)``
        is the second half; an agent that fuses the gather into its gradient step (``idqn_learn_on_replay``) takes the slots and
        ``ring_view()`` instead -- same keys from the same generator stream either way."""
        self.flush_deferred()
        assert self.add_count, ValueError("No samples in replay buffer!")
        if size is None:
            size = self._batch_size
        keys = self._sampling_distribution.sample(size)
        return (np.asarray(keys, np.int64) % self._max_capacity).astype(np.int32)

    def ring_view(self):
        """``(frames, n_frames, frame_bytes, rows, stack, frame_shape, dtype)`` of the device store, element rows flushed."""
        self._flush_meta()
        return self._frames, self._n_frames, self._frame_bytes, self._meta_dev, self._stack_size, self._frame_shape, self._obs_dtype

    def update(self, keys, **kwargs: Any) -> None:
        self.flush_deferred()
        self._sampling_distribution.update(keys, **kwargs)
