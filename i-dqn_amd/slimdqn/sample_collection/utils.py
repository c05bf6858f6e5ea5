"""Action selection and sample collection -- the other device call per environment step.

Mirrors the reference's ``slimdqn/sample_collection/utils.py:8-40``: epsilon-greedy over a key split three ways
(uniform draw, random action, kwargs key for ``best_action``), then one environment step into the replay buffer.
The greedy branch is one batch-1 forward of a head on the HIP path (``idqn_q_values``); ``.item()`` is the
reference's device->host sync (``utils.py:21``).
"""
from slimdqn import prng
from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement


class _HostAction(int):
    def item(self):
        return int(self)


def select_action(best_action_fn, params, state, key, n_actions, epsilon_fn, n_training_steps):
    uniform_key, action_key, kwargs_key = prng.split(key, 3)
    if prng.uniform(uniform_key) <= epsilon_fn(n_training_steps):
        return _HostAction(prng.randint(action_key, 0, n_actions))  # random action
    return best_action_fn(params, state, key=kwargs_key)  # greedy action (device scalar)


def linear_schedule(init_value: float, end_value: float, transition_steps):
    """optax.linear_schedule (experiments/base/dqn.py:19): linear from init to end over transition_steps, then flat."""

    def schedule(count):
        frac = 1.0 - min(max(count, 0), transition_steps) / transition_steps if transition_steps > 0 else 0.0
        return (init_value - end_value) * frac + end_value

    return schedule


def collect_single_sample(key, env, agent, rb: ReplayBuffer, p, epsilon_schedule, n_training_steps: int):
    action = select_action(agent.best_action, agent.params, env.state, key, env.n_actions, epsilon_schedule,
                           n_training_steps).item()
    obs = env.observation
    reward, absorbing = env.step(action)
    episode_end = absorbing or env.n_steps >= p["horizon"]
    rb.add(TransitionElement(observation=obs, action=action,
                             reward=reward if rb._clipping is None else rb._clipping(reward),
                             is_terminal=absorbing, episode_end=episode_end))
    if episode_end:
        env.reset()
    return reward, episode_end
