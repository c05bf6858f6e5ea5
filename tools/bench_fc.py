"""Step time of the MLP (LunarLander-style) i-DQN step and the latency of acting with it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from collections import namedtuple
from slimdqn.networks.idqn import iDQN
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
for K, feats in ((3, [100, 100]), (5, [200, 200]), (16, [100, 100])):
    agent = iDQN(0, 8, 4, K, feats, "fc", 3e-4, 0.99, 1, 1, 10**9, 10**9)
    rng = np.random.default_rng(0)
    b = Batch(torch.from_numpy(rng.standard_normal((32, 8)).astype(np.float32)).cuda(),
              torch.from_numpy(rng.integers(0, 4, 32).astype(np.int32)).cuda(),
              torch.from_numpy(rng.standard_normal(32).astype(np.float32)).cuda(),
              torch.from_numpy(rng.standard_normal((32, 8)).astype(np.float32)).cuda(),
              torch.from_numpy((rng.random(32) < 0.05).astype(np.uint8)).cuda())
    for _ in range(50): agent._learn(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 2000
    for _ in range(n): agent._learn(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"fc K={K} features={feats}: {dt*1e6:.1f} us/step = {1/dt:.0f} steps/s")

# acting path: greedy action of one state (idqn_best_action), synchronised like the trainer does with .item()
from slimdqn import prng
agent = iDQN(0, 8, 4, 3, [100, 100], "fc", 3e-4, 0.99, 1, 1, 10**9, 10**9)
state = np.zeros(8, np.float32)
key = prng.PRNGKey(0)
for _ in range(20): int(agent.best_action(agent.params, state, key))
t0 = time.perf_counter()
for _ in range(500): int(agent.best_action(agent.params, state, key))
print(f"fc best_action + .item(): {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per call")
