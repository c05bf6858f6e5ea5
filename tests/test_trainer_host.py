"""CPU: host logic of the "next" rows 8f-1 / 8f-2 -- epsilon-greedy selection, the epsilon schedule and the trainer's
call sequence (reference experiments/base/dqn.py:26-69, slimdqn/sample_collection/utils.py:8-40) with fake device objects."""
import numpy as np

from slimdqn import prng
from slimdqn.sample_collection.utils import linear_schedule, select_action


def test_linear_schedule_matches_optax_semantics():
    f = linear_schedule(1.0, 0.01, 1000)  # experiments/base/dqn.py:19
    assert f(0) == 1.0 and abs(f(500) - 0.505) < 1e-12 and f(1000) == 0.01 and f(5000) == 0.01


def test_select_action_epsilon_branches():
    class Greedy(int):
        def item(self):
            return int(self)

    calls = []

    def best(params, state, key):
        calls.append(key)
        return Greedy(3)

    key = prng.PRNGKey(0)
    assert select_action(best, None, None, key, 6, lambda n: 0.0, 0).item() == 3 and len(calls) == 1
    acts = {select_action(best, None, None, prng.PRNGKey(s), 6, lambda n: 1.0, 0).item() for s in range(60)}
    assert acts <= set(range(6)) and len(acts) > 1 and len(calls) == 1  # eps = 1: always random, greedy not evaluated
    a1 = select_action(best, None, None, prng.PRNGKey(7), 6, lambda n: 1.0, 0).item()
    assert a1 == select_action(best, None, None, prng.PRNGKey(7), 6, lambda n: 1.0, 0).item()  # deterministic per key


def test_train_call_sequence_with_fakes():
    from experiments.base.dqn import train
    from experiments.base.utils import NullLogger
    from slimdqn.environments.synthetic import SyntheticVector

    class FakeRB:
        _clipping = None

        def __init__(self):
            self.n = 0

        def add(self, tr):
            self.n += 1

    class FakeAgent:
        params = None

        def __init__(self):
            self.calls = []

        def best_action(self, params, state, key):
            class A(int):
                def item(self):
                    return int(self)
            return A(0)

        def update_online_params(self, step, rb):
            self.calls.append(("online", step))

        def update_target_params(self, step):
            self.calls.append(("target", step))
            return (step % 10 == 0), ({"loss": 1.0} if step % 10 == 0 else {})

        def get_model(self):
            return {"params": {}}

    p = dict(epsilon_end=0.01, epsilon_duration=10, n_epochs=2, n_training_steps_per_epoch=30, n_initial_samples=5,
             horizon=1000, wandb=NullLogger())
    agent, rb, env = FakeAgent(), FakeRB(), SyntheticVector(0, episode_length=7)
    returns, lengths = train(prng.PRNGKey(0), p, agent, env, rb)
    steps = [s for kind, s in agent.calls if kind == "online"]
    assert steps[0] == 6 and steps == list(range(6, 6 + len(steps)))  # starts after n_initial_samples
    assert [c for c in agent.calls[:2]] == [("online", 6), ("target", 6)]  # online then target, every step
    assert len(returns) == 2 and all(sum(l) >= 30 for l in lengths) and rb.n == sum(sum(l) for l in lengths)
    assert any("loss" in r for r in p["wandb"].records) and sum("epoch" in r for r in p["wandb"].records) == 2
