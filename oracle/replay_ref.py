"""CPU oracle (test infrastructure only): the reference's replay buffer semantics.

Restates ``slimdqn/sample_collection/replay_buffer.py:74-237`` of the reference without its
jax / flax / snappy dependencies (compression is a host-RAM trick with no observable effect:
``pack`` then ``unpack`` is the identity, ``tests/test_replay_buffer.py:21-49``):

* trajectory window of ``update_horizon + stack_size`` transitions (``:101``);
* ``_make_replay_element`` (``:103-180``): validity rule, effective horizon for a terminal that
  arrives before ``n`` steps, zero-padded frame stacks, action taken at the last frame of the
  ``state`` stack, n-step discounted reward over ``[stop, stop + n - 1]`` clipped to the window,
  ``episode_end := is_terminal`` of the last transition (``:146``);
* ``accumulate`` (``:182-200``): terminal flush loop; truncation (``episode_end``) clears;
* ``add`` (``:202-213``): keys are the running ``add_count``; FIFO eviction once
  ``add_count > max_capacity``;
* ``sample`` (``:215-230``): keys from the sampler, elements stacked along a new leading axis.
"""
import collections

import numpy as np

Element = collections.namedtuple("Element", "state action reward next_state is_terminal episode_end")
Transition = collections.namedtuple("Transition", "observation action reward is_terminal episode_end")
Transition.__new__.__defaults__ = (False,)


def element_from_window(window, stack_size, n, gamma):
    """One replay element from the current trajectory window, or None (replay_buffer.py:103-180)."""
    length = len(window)
    tail = window[-1]
    if not (length > n or (length > 1 and tail.is_terminal)):
        return None
    horizon = n
    if tail.is_terminal and length <= n:
        horizon = length - 1
    obs = np.asarray(tail.observation)
    shape = obs.shape + (stack_size,)
    state = np.zeros(shape, obs.dtype)
    nxt = np.zeros(shape, obs.dtype)
    s_lo, s_hi = length - horizon - stack_size, length - horizon - 1  # inclusive bounds
    n_lo, n_hi = length - stack_size, length - 1
    r_lo, r_hi = s_hi, s_hi + n - 1
    reward = 0.0
    for t, tr in enumerate(window):
        if r_lo <= t <= r_hi:
            reward += tr.reward * (gamma ** (t - r_lo))
        if s_lo <= t <= s_hi:
            state[..., t - s_lo] = tr.observation
        if n_lo <= t <= n_hi:
            nxt[..., t - n_lo] = tr.observation
    return Element(
        state=state,
        action=window[s_hi].action,
        reward=reward,
        next_state=nxt,
        is_terminal=window[n_hi].is_terminal,
        episode_end=window[n_hi].is_terminal,
    )


class ReplayRef:
    def __init__(self, sampling_distribution, batch_size, max_capacity, stack_size=4, update_horizon=1,
                 gamma=0.99, checkpoint_duration=4, compress=True, clipping=None):
        sampler = sampling_distribution  # compress / checkpoint_duration: no observable effect
        self.add_count = 0
        self.max_capacity = max_capacity
        self.memory = collections.OrderedDict()
        self.sampler = sampler
        self.batch_size = batch_size
        self.stack_size = stack_size
        self.n = update_horizon
        self.gamma = gamma
        self._clipping = clipping
        self.window = collections.deque(maxlen=update_horizon + stack_size)

    _memory = property(lambda self: self.memory)
    _sampling_distribution = property(lambda self: self.sampler)
    _max_capacity = property(lambda self: self.max_capacity)

    def accumulate(self, transition):
        self.window.append(transition)
        out = []
        if transition.is_terminal:
            while True:
                el = element_from_window(self.window, self.stack_size, self.n, self.gamma)
                if el is None:
                    break
                out.append(el)
                self.window.popleft()
            self.window.clear()
        else:
            el = element_from_window(self.window, self.stack_size, self.n, self.gamma)
            if el is not None:
                out.append(el)
            if transition.episode_end:
                self.window.clear()
        return out

    def add(self, transition, **kwargs):
        for el in self.accumulate(transition):
            key = self.add_count
            self.memory[key] = el
            self.sampler.add(key, **kwargs)
            self.add_count += 1
            if self.add_count > self.max_capacity:
                oldest, _ = self.memory.popitem(last=False)
                self.sampler.remove(oldest)

    def sample(self, size=None):
        assert self.add_count, ValueError("No samples in replay buffer!")
        size = self.batch_size if size is None else size
        keys = self.sampler.sample(size)
        els = [self.memory[int(k)] for k in keys]
        return Element(*[np.stack([getattr(e, f) for e in els]) for f in Element._fields])

    def update(self, keys, **kwargs):
        self.sampler.update(keys, **kwargs)
