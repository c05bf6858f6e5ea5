import os, sys, time
ROOT = os.getcwd(); sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from slimdqn import prng
from slimdqn.networks.idqn import iDQN
agent = iDQN(0, (84, 84, 4), 6, 5, [32, 64, 64, 512], "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
state = np.zeros((84, 84, 4), np.uint8); key = prng.PRNGKey(0)
for _ in range(200): int(agent.best_action(agent.params, state, key))
