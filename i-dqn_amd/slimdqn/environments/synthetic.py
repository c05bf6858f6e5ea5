"""Synthetic environments with the reference's environment protocol (``.reset() .step(a) -> (reward, absorbing)
.state .observation .n_steps .n_actions``, ``slimdqn/environments/atari.py:13-89``, ``lunar_lander.py:5-24``).

The real emulators (ALE, Box2D) are host-CPU code outside the hot path's scope and are not installed here; these
stand-ins produce Atari-shaped (84x84 uint8 frames, 4-stack state) or vector observations so that the trainer loop
can be driven end to end on the GPU box.
"""
import numpy as np


class SyntheticAtari:
    def __init__(self, seed=0, n_actions=6, episode_length=64):
        self.rng = np.random.default_rng(seed)
        self.n_actions, self.episode_length = n_actions, episode_length
        self.state_height, self.state_width, self.n_stacked_frames = 84, 84, 4
        # a pool of random frames: drawing 7056 fresh integers per step cost more host time than the rest of a step
        self._pool = self.rng.integers(0, 256, (256, 84, 84), dtype=np.uint8)
        self._rewards = self.rng.integers(-1, 2, 4096).astype(np.float64)
        # ... and the 4-stacks of consecutive pool frames, so that a step away from an episode start is a lookup (building
        # the channel-interleaved stack costs ~30 us of host time per step)
        idx = (np.arange(256)[:, None] + np.arange(-3, 1)[None, :]) % 256
        self._stacks = np.ascontiguousarray(self._pool[idx].transpose(0, 2, 3, 1))  # [256][84][84][4]
        self._stacks.setflags(write=False)  # `state` hands out views of it
        self._n = 0

    @property
    def observation(self):  # newest frame only; the replay accumulator rebuilds stacks (atari.py:40-41)
        return np.copy(self.state[:, :, -1])

    def reset(self):
        self.state = np.zeros((84, 84, 4), np.uint8)
        self.state[:, :, -1] = self._pool[self._n % 256]
        self._n += 1
        self.n_steps = 0

    def step(self, action):
        i = self._n % 256
        self._n += 1
        self.n_steps += 1
        if self.n_steps >= 3:  # the window holds pool frames i - 3 .. i only
            self.state = self._stacks[i]
        else:
            self.state = np.concatenate([self.state[:, :, 1:], self._pool[i][:, :, None]], axis=2)
        return float(self._rewards[self._n % 4096]), bool(self.n_steps >= self.episode_length)


class SyntheticVector:
    def __init__(self, seed=0, dim=8, n_actions=4, episode_length=50):
        self.rng = np.random.default_rng(seed)
        self.observation_shape, self.n_actions, self.episode_length = (dim,), n_actions, episode_length

    @property
    def observation(self):
        return np.copy(self.state)

    def reset(self):
        self.state = self.rng.standard_normal(self.observation_shape).astype(np.float32)
        self.n_steps = 0

    def step(self, action):
        self.state = self.rng.standard_normal(self.observation_shape).astype(np.float32)
        self.n_steps += 1
        return float(self.rng.normal()), bool(self.n_steps >= self.episode_length)
