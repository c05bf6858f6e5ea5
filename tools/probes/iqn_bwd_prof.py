"""Timeline of the merged i-IQN Dense_0 gradient launch with Adam in the weight gradient's epilogue (k_iqn_d0_bwd_adam): per
workgroup start / products done / gate passed / end on the 100 MHz clock, debug build with IDQN_CONV_PROF=11."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IDQN_HIP_LIB", os.path.join(root, "i-dqn_amd", "libidqn_hip_debug.so"))
os.environ["IDQN_CONV_PROF"] = "11"
sys.path[:0] = [root, os.path.join(root, "i-dqn_amd")]
from collections import namedtuple

import numpy as np
import torch

import bench
from slimdqn.networks.iiqn import iIQN

Batch = namedtuple("Batch", "state action reward next_state is_terminal")
K, N = bench.K_HEADS, 32
agent = iIQN(0, bench.OBS, bench.N_ACTIONS, K, bench.FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4, n_quantiles=N)
b = Batch(*(torch.from_numpy(x).cuda() for x in bench.synthetic(1000, bench.N_ACTIONS)))
for _ in range(6):
    agent._learn(b)
torch.cuda.synchronize()
nbg = N // 8
raw = agent._debug("cprof").cpu().numpy().view(np.int64)
raw = raw[: len(raw) // 8 * 8].reshape(-1, 8)
live = raw[:, 0] != 0
p = raw[:, :4]
within = raw[:, 4] & 255
t0 = p[live, 0].min()
us = lambda x: (x - t0) / 100.0
d = live & (within < nbg)
w = live & (within >= nbg)
print(f"{d.sum()} data-gradient items, {w.sum()} weight-gradient items; launch span {us(p[live, 3].max()):.1f} us")
print(f"data-gradient item: median {np.median(p[d, 3] - p[d, 0]) / 100:.1f} us (p10 {np.percentile(p[d, 3] - p[d, 0], 10) / 100:.1f}, p90 {np.percentile(p[d, 3] - p[d, 0], 90) / 100:.1f})")
mf, wt, ep = p[w, 1] - p[w, 0], p[w, 2] - p[w, 1], p[w, 3] - p[w, 2]
for nm, x in (("products", mf), ("gate wait", wt), ("adam epilogue", ep)):
    print(f"weight-gradient item, {nm}: median {np.median(x) / 100:.1f} us (p10 {np.percentile(x, 10) / 100:.1f}, p90 {np.percentile(x, 90) / 100:.1f}, max {x.max() / 100:.1f})")
ends = np.sort(us(p[live, 3]))
print("last item ends at", ends[-1], "; 90 % of the items have ended by", ends[int(0.9 * len(ends))], "; items ending in the last 50 us:", (ends > ends[-1] - 50).sum(),
      "of them weight-gradient:", (us(p[w, 3]) > ends[-1] - 50).sum())
starts = np.sort(us(p[live, 0]))
print("starts: first round (<1 us):", (starts < 1).sum(), " last start at", starts[-1], " last weight-gradient start at", us(p[w, 0]).max())
# occupancy of the epilogue over time: how many items are in their epilogue per 10 us bin
edges = np.arange(0, ends[-1] + 10, 10.0)
occ = [int(((us(p[w, 2]) < e + 10) & (us(p[w, 3]) > e)).sum()) for e in edges]
print("items in their epilogue per 10 us bin:", occ)
