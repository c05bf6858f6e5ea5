"""Acting and sample collection -- the other device call of an environment step.

Same entry points as the reference's ``slimdqn/sample_collection/utils.py:8-40`` (``select_action``,
``collect_single_sample``): the key is split three ways (exploration draw, random action, key handed to
``best_action``); the greedy branch is one batch-1 forward of a head on the HIP path (``idqn_best_action``) whose
``.item()`` is the device -> host sync the reference has at ``utils.py:21``.
"""
from slimdqn import prng
from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement


class HostAction(int):
    """A host-side action that answers ``.item()`` like the device scalar of the greedy branch."""

    def item(self):
        return int(self)


def linear_schedule(init_value: float, end_value: float, transition_steps):
    """optax.linear_schedule (experiments/base/dqn.py:19): linear from init to end over transition_steps, then flat."""
    span = init_value - end_value

    def value_at(count):
        if transition_steps <= 0:  # optax.linear_schedule: a constant schedule at init_value
            return init_value
        progress = min(max(count, 0), transition_steps) / transition_steps
        return end_value + span * (1.0 - progress)

    return value_at


def select_action(best_action_fn, params, state, key, n_actions, epsilon_fn, n_training_steps):
    explore_key, random_action_key, greedy_key = prng.split(key, 3)
    exploring = prng.uniform(explore_key) <= epsilon_fn(n_training_steps)
    if exploring:
        return HostAction(prng.randint(random_action_key, 0, n_actions))
    return best_action_fn(params, state, key=greedy_key)  # device scalar


def collect_single_sample(key, env, agent, rb: ReplayBuffer, p, epsilon_schedule, n_training_steps: int):
    """One environment step into the replay buffer; returns ``(reward, episode_ended)``.

    ``p["overlap_replay_add"]`` (set by this build's launcher, absent = off): the greedy action is launched without waiting
    (``DeviceAgent.lazy_host_actions``), the replay bookkeeping of the PREVIOUS transition runs while the GPU computes it,
    and this step's ``rb.add`` is postponed the same way (``add_deferred``: flushed before anything reads the buffer) --
    same transitions in the same order, ~20 us of host time per step moved under the ~40 us of the acting launch."""
    overlap = bool(p.get("overlap_replay_add", False)) and hasattr(rb, "add_deferred")
    observation = env.observation  # the frame BEFORE the action goes with it (utils.py:27-35)
    pending = select_action(agent.best_action, agent.params, env.state, key, env.n_actions, epsilon_schedule, n_training_steps)
    if overlap:
        rb.flush_deferred()
    action = pending.item()
    reward, absorbing = env.step(action)
    ended = bool(absorbing) or env.n_steps >= p["horizon"]
    stored_reward = rb._clipping(reward) if rb._clipping is not None else reward
    (rb.add_deferred if overlap else rb.add)(TransitionElement(observation, action, stored_reward, absorbing, ended))
    if ended:
        env.reset()
    return reward, ended
