"""End-to-end rate of the trainer loop on the synthetic environments (what a user of experiments/*/idqn.py sees):
environment step + select_action + replay add + (every update_to_data steps) sample + learn + target updates."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from slimdqn import prng

def cnn_acting_latency():
    from slimdqn.networks.idqn import iDQN
    agent = iDQN(0, (84, 84, 4), 6, 5, [32, 64, 64, 512], "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
    state = np.zeros((84, 84, 4), np.uint8)
    key = prng.PRNGKey(0)
    for _ in range(20): int(agent.best_action(agent.params, state, key))
    t0 = time.perf_counter()
    for _ in range(300): int(agent.best_action(agent.params, state, key))
    print(f"cnn best_action + .item(): {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call")

def loop(env_name, argv, steps):
    """Two figures: the whole `run` (what the round-1..3 lines reported: agent construction, parameter initialisation, the
    pickle of the final model and the first-launch warm-up are ~0.15 s of it, a third of a 2000-step run) and the loop alone
    (perf_counter around Trainer.run_epoch: what a run of millions of steps sees)."""
    import tempfile
    from experiments.base import dqn as D
    if env_name == "atari":
        from experiments.atari.idqn import run
    else:
        from experiments.lunar_lander.idqn import run
    inner = {"t": 0.0, "n": 0}
    orig = D.Trainer.run_epoch

    def timed(self, index):
        torch.cuda.synchronize()
        n0, t0 = self.total_steps, time.perf_counter()
        orig(self, index)
        torch.cuda.synchronize()
        inner["t"] += time.perf_counter() - t0
        inner["n"] += self.total_steps - n0

    D.Trainer.run_epoch = timed
    try:
        with tempfile.TemporaryDirectory() as d:
            t0 = time.perf_counter()
            run(argv, save_root=d)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    finally:
        D.Trainer.run_epoch = orig
    n = inner["n"]
    print(f"{env_name}: whole run {n} env steps in {dt:.2f} s = {n / dt:.0f} env steps/s ({dt / n * 1e6:.0f} us per env step); "
          f"loop alone {n / inner['t']:.0f} env steps/s ({inner['t'] / n * 1e6:.0f} us per env step)")

cnn_acting_latency()
loop("lunar_lander", ["-en", "b", "-s", "1", "-ne", "1", "-ntspe", "20000", "-nis", "200", "-rbc", "10000", "-nn", "3",
                      "-tuf", "200", "-tsf", "10", "-f", "100", "100", "-horizon", "200"], 20000)
loop("atari", ["-en", "b", "-s", "1", "-ne", "1", "-ntspe", "10000", "-nis", "200", "-rbc", "5000", "-nn", "5", "-at", "cnn",
               "-tuf", "200", "-tsf", "10", "-f", "32", "64", "64", "512", "-horizon", "200", "-bs", "32", "-utd", "4"], 10000)
