"""Trainer loop of the HIP build.

Interface and event order follow the reference's ``experiments/base/dqn.py:12-69`` -- per environment step: split the
key, collect one sample, and once ``n_initial_samples`` are in, ``update_online_params`` then
``update_target_params``; a log record on every target update and at every epoch end; the model saved per epoch -- but
the loop is organised as a small state machine (``EpochStats`` + ``Trainer``) instead of one nested loop.

``p["wandb"]`` only needs ``.log(dict)``; ``experiments.base.utils.NullLogger`` stands in when wandb is absent.
"""
import numpy as np

from slimdqn import prng
from slimdqn.sample_collection.utils import collect_single_sample, linear_schedule


class EpochStats:
    """Returns and lengths of the episodes of one epoch; the last entry is the episode in progress."""

    def __init__(self):
        self.returns, self.lengths = [0], [0]

    def record(self, reward):
        self.returns[-1] += reward
        self.lengths[-1] += 1

    def open_episode(self):
        self.returns.append(0)
        self.lengths.append(0)

    def summary(self):
        return float(np.mean(self.returns)), float(np.mean(self.lengths)), len(self.lengths)


class Trainer:
    def __init__(self, key, p, agent, env, rb, save_fn=None):
        self.key, self.p, self.agent, self.env, self.rb, self.save_fn = key, p, agent, env, rb, save_fn
        self.epsilon = linear_schedule(1.0, p["epsilon_end"], p["epsilon_duration"])
        self.total_steps = 0
        self.history = []  # one EpochStats per epoch

    def _environment_step(self):
        self.key, explore_key = prng.split(self.key)
        return collect_single_sample(explore_key, self.env, self.agent, self.rb, self.p, self.epsilon, self.total_steps)

    def _gradient_step(self):
        self.agent.update_online_params(self.total_steps, self.rb)
        updated, logs = self.agent.update_target_params(self.total_steps)
        if updated:
            self.p["wandb"].log({"n_training_steps": self.total_steps, **logs})

    def run_epoch(self, index):
        stats, budget, done, just_reset = EpochStats(), self.p["n_training_steps_per_epoch"], 0, False
        self.history.append(stats)
        # an epoch always ends on an episode boundary (experiments/base/dqn.py:29)
        while done < budget or not just_reset:
            reward, just_reset = self._environment_step()
            done += 1
            self.total_steps += 1
            stats.record(reward)
            if just_reset and done < budget:
                stats.open_episode()
            if self.total_steps > self.p["n_initial_samples"]:
                self._gradient_step()
        if hasattr(self.rb, "flush_deferred"):
            self.rb.flush_deferred()  # (a postponed add of the last step: the buffer is complete at every epoch end)
        avg_return, avg_length, n_episodes = stats.summary()
        print(f"\nEpoch {index}: Return {avg_return} averaged on {n_episodes} episodes.\n", flush=True)
        self.p["wandb"].log({"epoch": index, "n_training_steps": self.total_steps, "avg_return": avg_return,
                             "avg_length_episode": avg_length})
        if self.save_fn is not None:
            self.save_fn(self.p, [s.returns for s in self.history], [s.lengths for s in self.history],
                         self.agent.get_model())

    def run(self):
        self.env.reset()
        for index in range(self.p["n_epochs"]):
            self.run_epoch(index)
        return [s.returns for s in self.history], [s.lengths for s in self.history]


def train(key, p: dict, agent, env, rb, save_fn=None):
    """Same call as the reference's ``train`` (``experiments/base/dqn.py:12``); returns (returns, lengths) per epoch."""
    return Trainer(key, p, agent, env, rb, save_fn).run()
