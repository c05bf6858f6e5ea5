"""Gradient-step time of the Atari net against the number of heads K (K = 1 is plain DQN)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from collections import namedtuple
import bench
from slimdqn.networks.idqn import iDQN
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
b = Batch(*(torch.from_numpy(x).cuda() for x in bench.synthetic(0)))
for K in (1, 2, 3, 5, 8, 16, 32, 64):
    agent = iDQN(0, bench.OBS, bench.N_ACTIONS, K, bench.FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
    for _ in range(30): agent._learn(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for _ in range(n): agent._learn(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"K={K:3d}: {dt*1e6:8.1f} us/step  {1/dt:8.0f} steps/s  {K/dt:9.0f} head-steps/s", flush=True)
    del agent; torch.cuda.empty_cache()
