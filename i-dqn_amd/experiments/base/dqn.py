"""Trainer loop with the reference's call sequence (``experiments/base/dqn.py:12-69``): per environment step
split the key, collect one sample, and once ``n_initial_samples`` are in, ``update_online_params`` then
``update_target_params``; log on target updates and per epoch; save the model per epoch.

``p["wandb"]`` only needs ``.log(dict)``; ``experiments.base.utils.NullLogger`` stands in when wandb is absent.
"""
import numpy as np

from slimdqn import prng
from slimdqn.sample_collection.utils import collect_single_sample, linear_schedule


def train(key, p: dict, agent, env, rb, save_fn=None):
    epsilon_schedule = linear_schedule(1.0, p["epsilon_end"], p["epsilon_duration"])
    n_training_steps = 0
    env.reset()
    episode_returns_per_epoch, episode_lengths_per_epoch = [[0]], [[0]]
    for idx_epoch in range(p["n_epochs"]):
        n_training_steps_epoch, has_reset = 0, False
        while n_training_steps_epoch < p["n_training_steps_per_epoch"] or not has_reset:
            key, exploration_key = prng.split(key)
            reward, has_reset = collect_single_sample(exploration_key, env, agent, rb, p, epsilon_schedule,
                                                      n_training_steps)
            n_training_steps_epoch += 1
            n_training_steps += 1
            episode_returns_per_epoch[idx_epoch][-1] += reward
            episode_lengths_per_epoch[idx_epoch][-1] += 1
            if has_reset and n_training_steps_epoch < p["n_training_steps_per_epoch"]:
                episode_returns_per_epoch[idx_epoch].append(0)
                episode_lengths_per_epoch[idx_epoch].append(0)
            if n_training_steps > p["n_initial_samples"]:
                agent.update_online_params(n_training_steps, rb)
                target_updated, logs = agent.update_target_params(n_training_steps)
                if target_updated:
                    p["wandb"].log({"n_training_steps": n_training_steps, **logs})
        avg_return = np.mean(episode_returns_per_epoch[idx_epoch])
        avg_length_episode = np.mean(episode_lengths_per_epoch[idx_epoch])
        print(f"\nEpoch {idx_epoch}: Return {avg_return} averaged on "
              f"{len(episode_lengths_per_epoch[idx_epoch])} episodes.\n", flush=True)
        p["wandb"].log({"epoch": idx_epoch, "n_training_steps": n_training_steps, "avg_return": avg_return,
                        "avg_length_episode": avg_length_episode})
        if idx_epoch < p["n_epochs"] - 1:
            episode_returns_per_epoch.append([0])
            episode_lengths_per_epoch.append([0])
        if save_fn is not None:
            save_fn(p, episode_returns_per_epoch, episode_lengths_per_epoch, agent.get_model())
    return episode_returns_per_epoch, episode_lengths_per_epoch
