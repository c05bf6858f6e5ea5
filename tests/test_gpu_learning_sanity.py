"""GPU: does the whole loop LEARN?  Parity tests pin the arithmetic of one step; this drives the reference's call
sequence (select_action -> env.step -> replay add -> update_online_params -> update_target_params, with the i-DQN
shift / sync of the chain) on a task whose optimum is known, and checks that every head finds it.

Task: two-step episodes; the observation is a noisy one-hot of the rewarded action (4 actions, 8 dims), reward 1 for
that action and 0 otherwise, absorbing after the second step.  The replay element built from the two transitions
carries the first step's reward and the second transition's terminal flag (the reference's pairing,
replay_buffer.py:103-180), so Q*(s, a) = 1[a == target(s)] and a trained head acts greedily right.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class OneStepBandit:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.observation_shape, self.n_actions = (8,), 4
        self.n_steps = 0

    def _draw(self):
        self.target = int(self.rng.integers(4))
        s = 0.1 * self.rng.standard_normal(8).astype(np.float32)
        s[self.target] += 1.0
        self.state = s

    @property
    def observation(self):
        return np.copy(self.state)

    def reset(self):
        self._draw()
        self.n_steps = 0

    def step(self, action):
        reward = 1.0 if int(action) == self.target else 0.0
        self.n_steps += 1
        self._draw()
        return reward, self.n_steps >= 2


@pytest.mark.parametrize("algo", ["idqn", "dqn"])
def test_agents_learn_the_one_step_task(algo):
    from experiments.base.dqn import train
    from experiments.base.utils import NullLogger
    from slimdqn import prng
    from slimdqn.networks.dqn import DQN
    from slimdqn.networks.idqn import iDQN
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    p = {"epsilon_end": 0.05, "epsilon_duration": 1500, "n_epochs": 1, "n_training_steps_per_epoch": 4000,
         "n_initial_samples": 200, "horizon": 10, "wandb": NullLogger()}
    env = OneStepBandit(0)
    rb = ReplayBuffer(UniformSamplingDistribution(0), batch_size=32, max_capacity=5000, stack_size=1, update_horizon=1,
                      gamma=0.99)
    q_key, train_key = prng.split(prng.PRNGKey(0))
    common = dict(features=[64, 64], architecture_type="fc", learning_rate=1e-3, gamma=0.99, update_horizon=1,
                  update_to_data=1, target_update_frequency=100)
    if algo == "idqn":
        agent = iDQN(q_key, 8, 4, n_networks=3, target_sync_frequency=10, **common)
    else:
        agent = DQN(q_key, 8, 4, **common)
    train(train_key, p, agent, env, rb)
    # greedy accuracy of every head on fresh states
    test_env, K = OneStepBandit(123), getattr(agent, "n_networks", 1)
    hits = np.zeros(K)
    for _ in range(200):
        test_env.reset()
        for k in range(K):
            q = agent.q_values(agent.params, test_env.state, k) if algo == "idqn" else agent.q_values(agent.params, test_env.state)
            hits[k] += int(q[0].argmax().item()) == test_env.target
    assert (hits / 200 >= 0.95).all(), hits / 200
    logs = [r for r in p["wandb"].records if "loss" in r]
    assert logs and logs[-1]["loss"] < logs[0]["loss"]  # the TD loss went down over the run


class QuadrantFrames:
    """Atari-protocol environment (``state`` = stack of 4 frames, ``observation`` = newest frame): the newest frame
    has one bright quadrant; the rewarded action is that quadrant's index.  Two-step episodes as above."""

    def __init__(self, seed, size=20):
        self.rng = np.random.default_rng(seed)
        self.size, self.n_actions = size, 4
        self.state_height = self.state_width = size
        self.n_stacked_frames = 4
        self.n_steps = 0

    def _frame(self):
        self.target = int(self.rng.integers(4))
        f = self.rng.integers(0, 40, (self.size, self.size), dtype=np.uint8)
        h = self.size // 2
        r, c = divmod(self.target, 2)
        f[r * h : (r + 1) * h, c * h : (c + 1) * h] += 200
        return f

    @property
    def observation(self):
        return np.copy(self.state[:, :, -1])

    def reset(self):
        self.state = np.zeros((self.size, self.size, 4), np.uint8)
        self.state[:, :, -1] = self._frame()
        self.n_steps = 0

    def step(self, action):
        reward = 1.0 if int(action) == self.target else 0.0
        self.n_steps += 1
        self.state = np.concatenate([self.state[:, :, 1:], self._frame()[:, :, None]], axis=2)
        return reward, self.n_steps >= 2


@pytest.mark.parametrize("conv", ["f32", "bf16x3"])
def test_cnn_heads_learn_from_pixels(conv, monkeypatch):
    from experiments.base.dqn import train
    from experiments.base.utils import NullLogger
    from slimdqn import prng
    from slimdqn.networks.idqn import iDQN
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    monkeypatch.setenv("IDQN_CONV", conv)
    p = {"epsilon_end": 0.05, "epsilon_duration": 1000, "n_epochs": 1, "n_training_steps_per_epoch": 3000,
         "n_initial_samples": 200, "horizon": 10, "wandb": NullLogger()}
    env = QuadrantFrames(0)
    rb = ReplayBuffer(UniformSamplingDistribution(0), batch_size=32, max_capacity=4000, stack_size=4, update_horizon=1,
                      gamma=0.99)
    q_key, train_key = prng.split(prng.PRNGKey(1))
    agent = iDQN(q_key, (20, 20, 4), 4, n_networks=2, features=[32, 32, 32, 128], architecture_type="cnn",
                 learning_rate=3e-4, gamma=0.99, update_horizon=1, update_to_data=1, target_update_frequency=100,
                 target_sync_frequency=10, adam_eps=1.5e-4)
    train(train_key, p, agent, env, rb)
    test_env, hits = QuadrantFrames(77), np.zeros(2)
    for _ in range(100):
        test_env.reset()
        test_env.step(0)  # a state with two real frames
        for k in range(2):
            hits[k] += int(agent.q_values(agent.params, test_env.state, k)[0].argmax().item()) == test_env.target
    assert (hits / 100 >= 0.9).all(), hits / 100
