"""Atari DQN entry point with the reference's wiring (``experiments/atari/dqn.py:15-49``) -- the script the
reference's own integration test runs (``tests/test_atari.py:15-59``): uniform sampler seeded by the experiment seed,
reward clipping, stack 4, ``adam_eps = 1.5e-4``.  ``env`` defaults to the synthetic Atari-shaped environment."""
import sys

import numpy as np

from experiments.base.dqn import train
from experiments.base.utils import prepare_logs, save_data
from slimdqn import prng
from slimdqn.networks.dqn import DQN
from slimdqn.sample_collection.replay_buffer import ReplayBuffer
from slimdqn.sample_collection.samplers import UniformSamplingDistribution


def run(argvs=sys.argv[1:], env=None, save_root=None):
    p = prepare_logs("atari", "dqn", argvs, save_root)
    q_key, train_key = prng.split(prng.PRNGKey(p["seed"]))
    if env is None:
        from slimdqn.environments.synthetic import SyntheticAtari

        env = SyntheticAtari(p["seed"])
    rb = ReplayBuffer(sampling_distribution=UniformSamplingDistribution(p["seed"]),
                      max_capacity=p["replay_buffer_capacity"], batch_size=p["batch_size"],
                      update_horizon=p["update_horizon"], gamma=p["gamma"], clipping=lambda x: np.clip(x, -1, 1),
                      stack_size=4, compress=True)
    agent = DQN(q_key, (env.state_height, env.state_width, env.n_stacked_frames), env.n_actions,
                features=p["features"], architecture_type=p["architecture_type"], learning_rate=p["learning_rate"],
                gamma=p["gamma"], update_horizon=p["update_horizon"], update_to_data=p["update_to_data"],
                target_update_frequency=p["target_update_frequency"], adam_eps=1.5e-4)
    train(train_key, p, agent, env, rb, save_fn=save_data)
    return p, agent


if __name__ == "__main__":
    run()
