"""Phase timestamps of the Conv_2 forward launch of the bf16x3 path (IDQN_CONV=bf16x3 IDQN_CONV_PROF=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from collections import namedtuple
from slimdqn.networks.idqn import iDQN
import bench
agent = iDQN(0, (84, 84, 4), 6, 5, [32, 64, 64, 512], "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
b = Batch(*(torch.from_numpy(x).cuda() for x in bench.synthetic(0)))
for _ in range(20):
    agent._learn(b)
torch.cuda.synchronize()
p = agent._debug("c3prof").cpu().numpy().view(np.int64).reshape(-1, 4)[:610].copy()
place = (p[:, 2] >> 48) & 0xfff  # (xcc << 8) | (se, sh, cu)
p[:, 2] &= (1 << 48) - 1
t0 = p[:, 0].min()
q = (p - t0) / 100.0  # wall_clock64 ticks at 100 MHz -> microseconds
print("WG start   us: min %.2f median %.2f max %.2f" % (q[:, 0].min(), np.median(q[:, 0]), q[:, 0].max()))
print("prologue   us: median %.2f max %.2f" % (np.median(q[:, 1] - q[:, 0]), (q[:, 1] - q[:, 0]).max()))
print("k-loop     us: median %.2f max %.2f" % (np.median(q[:, 2] - q[:, 1]), (q[:, 2] - q[:, 1]).max()))
cyc = p[:, 3].astype(np.float64)
loop_us = (p[:, 2] - p[:, 1]) / 100.0
print("k-loop shader cycles: median %.0f ; cycles per microsecond (= MHz): median %.0f min %.0f max %.0f" % (np.median(cyc), np.median(cyc / loop_us), (cyc / loop_us).min(), (cyc / loop_us).max()))
order = np.argsort(q[:, 0])
print("start times of every 50th WG:", np.round(q[order[::50], 0], 1))

import collections
per_cu = collections.Counter(place.tolist())
hist = collections.Counter(per_cu.values())
print("distinct (xcc, se, sh, cu) slots used: %d ; workgroups per slot histogram: %s" % (len(per_cu), dict(sorted(hist.items()))))
print("workgroups per XCC:", dict(sorted(collections.Counter((place >> 8).tolist()).items())))
