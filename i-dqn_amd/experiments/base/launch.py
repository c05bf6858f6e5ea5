"""One launcher for the entry points (``experiments/{atari,lunar_lander}/{dqn,idqn}.py`` of the reference, plus
``experiments/atari/iiqn.py`` for the quantile-head extension: they wire
the same four objects -- flags, environment, replay buffer, agent -- with per-environment constants; here that wiring
is a table and a function instead of four near-identical scripts)."""
import os

import numpy as np

from experiments.base.dqn import train
from experiments.base.utils import prepare_logs, save_data
from slimdqn import prng
from slimdqn.sample_collection.replay_buffer import ReplayBuffer
from slimdqn.sample_collection.samplers import UniformSamplingDistribution

# per-environment constants of the reference's scripts: frame stack, reward clipping, Adam epsilon
ENVIRONMENTS = {
    "atari": dict(stack_size=4, clipping=lambda r: np.clip(r, -1, 1), adam_eps=1.5e-4),   # atari/idqn.py:24-46
    "lunar_lander": dict(stack_size=1, clipping=None, adam_eps=1e-8),                     # lunar_lander/idqn.py:23-43
}


def default_environment(env_name, seed):
    from slimdqn.environments import synthetic

    return synthetic.SyntheticAtari(seed) if env_name == "atari" else synthetic.SyntheticVector(seed)


def observation_dim(env_name, env):
    if env_name == "atari":
        return (env.state_height, env.state_width, env.n_stacked_frames)
    return env.observation_shape[0]  # the reference passes an int for LunarLander


def make_agent(algo, key, obs_dim, n_actions, p, adam_eps):
    shared = dict(features=p["features"], architecture_type=p["architecture_type"], learning_rate=p["learning_rate"],
                  gamma=p["gamma"], update_horizon=p["update_horizon"], update_to_data=p["update_to_data"],
                  target_update_frequency=p["target_update_frequency"], adam_eps=adam_eps)
    if algo == "idqn":
        from slimdqn.networks.idqn import iDQN

        return iDQN(key, obs_dim, n_actions, n_networks=p["n_networks"],
                    target_sync_frequency=p["target_sync_frequency"], **shared)
    if algo == "iiqn":  # extension: the quantile heads on the same trunk and chain (slimdqn/networks/iiqn.py)
        from slimdqn.networks.iiqn import iIQN

        return iIQN(key, obs_dim, n_actions, n_networks=p["n_networks"],
                    target_sync_frequency=p["target_sync_frequency"], n_quantiles=p["n_quantiles"], **shared)
    from slimdqn.networks.dqn import DQN

    return DQN(key, obs_dim, n_actions, **shared)


def launch(env_name, algo, argvs, env=None, save_root=None):
    """Parses the flags, builds environment / replay buffer / agent and trains; returns ``(p, agent)``."""
    # the trainer steps the same two batch-buffer sets over and over: let the library replay the step's launches as one
    # hipGraph per set (IDQN_STEP_GRAPH, default 0 everywhere else, is switched ON here; read once, at the first step;
    # results are bit-identical: tests/test_gpu_switches.py runs this entry point both ways)
    os.environ.setdefault("IDQN_STEP_GRAPH", "1")
    consts = ENVIRONMENTS[env_name]
    p = prepare_logs(env_name, algo, argvs, save_root)
    agent_key, train_key = prng.split(prng.PRNGKey(p["seed"]))
    env = env if env is not None else default_environment(env_name, p["seed"])
    rb = ReplayBuffer(UniformSamplingDistribution(p["seed"]), batch_size=p["batch_size"],
                      max_capacity=p["replay_buffer_capacity"], stack_size=consts["stack_size"],
                      update_horizon=p["update_horizon"], gamma=p["gamma"], clipping=consts["clipping"], compress=True)
    rb.reuse_sample_buffers = True  # the loop below consumes every batch before it draws the next (two staging sets in turn)
    agent = make_agent(algo, agent_key, observation_dim(env_name, env), env.n_actions, p, consts["adam_eps"])
    # host work under the acting launch (slimdqn/sample_collection/utils.py, collect_single_sample); IDQN_LOOP_OVERLAP=0: off.
    # Only where the launch is long enough to hide something: the conv nets' ~40 us (Atari-shaped loop 6.15 -> 6.5 k env
    # steps/s); the MLP's ~20 us launch is shorter than what the extra calls cost (7.3 -> 7.0 k), so it stays synchronous.
    if os.environ.get("IDQN_LOOP_OVERLAP", "1") != "0" and hasattr(agent, "lazy_host_actions") and p["architecture_type"] == "cnn":
        agent.lazy_host_actions = True
        p["overlap_replay_add"] = True
    train(train_key, p, agent, env, rb, save_fn=save_data)
    return p, agent
