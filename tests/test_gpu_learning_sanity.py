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
