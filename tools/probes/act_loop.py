"""Acting path only: 300 calls of iDQN.best_action on one Atari-shaped state (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np
from slimdqn import prng
from slimdqn.networks.idqn import iDQN
agent = iDQN(0, (84, 84, 4), 6, 5, [32, 64, 64, 512], "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
state = np.random.default_rng(0).integers(0, 256, (84, 84, 4), dtype=np.uint8)
key = prng.PRNGKey(0)
for _ in range(20): int(agent.best_action(agent.params, state, key))
t0 = time.perf_counter()
for _ in range(300): int(agent.best_action(agent.params, state, key))
print(f"cnn best_action + .item(): {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call")
