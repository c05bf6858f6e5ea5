"""Timeline of the chained forward conv launch (IDQN_CONV_PROF=10): per layer, when its items start and end relative to the
launch (100 MHz wall clock), the shader cycles of each phase, and what an item spends polling its producers' flags."""
import os
os.environ.setdefault("IDQN_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "i-dqn_amd", "libidqn_hip_variants.so"))  # variants build: the stamps / switches used here
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
os.environ["IDQN_CONV_PROF"] = "10"
os.environ["IDQN_CONV_CHAIN"] = "1"  # (the chained launch is opt-in)
import numpy as np
import torch
from collections import namedtuple

import bench
from slimdqn.networks.idqn import iDQN

agent = iDQN(0, bench.OBS, bench.N_ACTIONS, bench.K_HEADS, bench.FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
b = Batch(*(torch.from_numpy(x).cuda() for x in bench.synthetic(1)))
for _ in range(30):
    agent._learn(b)
torch.cuda.synchronize()
allp = agent._debug("cprof").cpu().numpy().view(np.int64).reshape(3, 2, 4096, 8)
t0 = min(allp[i, 0][allp[i, 0][:, 0] != 0][:, 0].min() for i in range(3) if (allp[i, 0][:, 0] != 0).any())
q = lambda c: f"median {np.median(c):8.1f}  p10 {np.percentile(c, 10):8.1f}  p90 {np.percentile(c, 90):8.1f}"
for i in range(3):
    raw, ld = allp[i, 0], allp[i, 1]
    keep = raw[:, 0] != 0
    raw, ld = raw[keep], ld[keep]
    if not len(raw):
        print(f"layer {i}: no stamps (launches not chained?)")
        continue
    print(f"Conv_{i}: {len(raw)} items; starts {q((raw[:, 0] - t0) / 100)} us; ends {q((raw[:, 6] - t0) / 100)} us; last end {(raw[:, 6].max() - t0) / 100:.2f} us")
    for j, n in enumerate(["prologue", "first fill", "loop", "epilogue", "  waits in loop"]):
        print(f"    {n:16s} {q(raw[:, 1 + j])} cycles")
    if i > 0:
        print(f"    poll + acquire   {q(ld[:, 4])} cycles; flags ready at {q((ld[:, 5] - t0) / 100)} us; flags per item {np.unique(ld[:, 6])}")
