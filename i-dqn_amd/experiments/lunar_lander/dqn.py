"""LunarLander DQN entry point with the reference's wiring (``experiments/lunar_lander/dqn.py:15-46``; run by
``tests/test_lunar_lander.py``): stack size 1, integer ``observation_dim``, default ``adam_eps``.  ``env`` defaults to
a synthetic 8-dim vector environment."""
import sys

from experiments.base.dqn import train
from experiments.base.utils import prepare_logs, save_data
from slimdqn import prng
from slimdqn.networks.dqn import DQN
from slimdqn.sample_collection.replay_buffer import ReplayBuffer
from slimdqn.sample_collection.samplers import UniformSamplingDistribution


def run(argvs=sys.argv[1:], env=None, save_root=None):
    p = prepare_logs("lunar_lander", "dqn", argvs, save_root)
    q_key, train_key = prng.split(prng.PRNGKey(p["seed"]))
    if env is None:
        from slimdqn.environments.synthetic import SyntheticVector

        env = SyntheticVector(p["seed"])
    rb = ReplayBuffer(sampling_distribution=UniformSamplingDistribution(p["seed"]), batch_size=p["batch_size"],
                      max_capacity=p["replay_buffer_capacity"], stack_size=1, update_horizon=p["update_horizon"],
                      gamma=p["gamma"], compress=True)
    agent = DQN(q_key, env.observation_shape[0], env.n_actions, features=p["features"],
                architecture_type=p["architecture_type"], learning_rate=p["learning_rate"], gamma=p["gamma"],
                update_horizon=p["update_horizon"], update_to_data=p["update_to_data"],
                target_update_frequency=p["target_update_frequency"])
    train(train_key, p, agent, env, rb, save_fn=save_data)
    return p, agent


if __name__ == "__main__":
    run()
