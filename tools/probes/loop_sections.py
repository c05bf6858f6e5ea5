"""Where the wall time of an Atari-shaped environment step goes (no cProfile: perf_counter accumulators around the pieces
of the trainer loop, experiments/base/dqn.py), and how much of it is the host waiting for the GPU inside best_action."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import torch
from collections import defaultdict

acc, cnt = defaultdict(float), defaultdict(int)


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0
            cnt[label] += 1

    setattr(obj, name, timed)


from slimdqn import prng
from slimdqn.networks import _agent
from slimdqn.sample_collection import replay_buffer as RB, utils as U
from experiments.base import dqn as D

wrap(_agent.DeviceAgent, "_best_action", "best_action (C call: launch + wait for the GPU)")
wrap(_agent.DeviceAgent, "_learn", "learn_on_batch host side")
wrap(RB.ReplayBuffer, "add", "replay add")
wrap(RB.ReplayBuffer, "sample", "replay sample (indices, slots, gather launch)")
wrap(D.Trainer, "_gradient_step", "gradient step, all host work")
wrap(D.Trainer, "_environment_step", "environment step, all")
wrap(prng, "split", "prng.split")
from experiments.atari.idqn import run

argv = ["-en", "b", "-s", "1", "-ne", "1", "-ntspe", "3000", "-nis", "200", "-rbc", "5000", "-nn", "5", "-at", "cnn",
        "-tuf", "200", "-tsf", "10", "-f", "32", "64", "64", "512", "-horizon", "200", "-bs", "32", "-utd", "4"]
with tempfile.TemporaryDirectory() as d:
    run(argv[:7] + ["300"] + argv[8:], save_root=d)
    acc.clear(); cnt.clear()
    t0 = time.perf_counter()
    run(argv, save_root=d)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
n = cnt["environment step, all"]
print(f"{n} env steps, {wall / n * 1e6:.1f} us per env step wall ({n / wall:.0f} env steps/s)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:58s} {v / n * 1e6:7.1f} us per env step  ({cnt[k]} calls, {v / cnt[k] * 1e6:7.1f} us per call)")
