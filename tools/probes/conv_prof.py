"""Phase timing inside one plane conv launch (IDQN_CONV_PROF=role): medians over workgroups of the shader-clock cycles
spent in the prologue, the first fill, the superstep loop (and waiting inside it) and the epilogue."""
import os
os.environ.setdefault("IDQN_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "i-dqn_amd", "libidqn_hip_debug.so"))  # debug build (__graft_entry__.build_debug()): the stamps / switches used here
import os
import sys

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "i-dqn_amd")]
import numpy as np
import torch
from collections import namedtuple

import bench
from slimdqn.networks.idqn import iDQN

role = int(os.environ["IDQN_CONV_PROF"])
KH_, B_ = int(os.environ.get("CPROF_HEADS", bench.K_HEADS)), int(os.environ.get("CPROF_BATCH", bench.BATCH))
agent = iDQN(0, bench.OBS, bench.N_ACTIONS, KH_, bench.FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
b = Batch(*(torch.from_numpy(x).cuda() for x in bench.synthetic(1, batch=B_)))
for _ in range(30):
    agent._learn(b)
torch.cuda.synchronize()
allp = agent._debug("cprof").cpu().numpy().view(np.int64).reshape(2, -1, 8)
raw, ld = allp[0], allp[1]
ld = ld[raw[:, 0] != 0]
raw = raw[raw[:, 0] != 0]
t0 = raw[:, 0].min()
print(f"role {role}: {len(raw)} workgroups; launch span {(raw[:, 6].max() - t0) / 100:.2f} us (wall), starts spread {(raw[:, 0].max() - t0) / 100:.2f} us")
names = ["prologue", "first fill", "loop", "epilogue", "  waits in loop"]
for i, n in enumerate(names):
    c = raw[:, 1 + i]
    print(f"  {n:16s} median {np.median(c):8.0f}  p10 {np.percentile(c, 10):8.0f}  p90 {np.percentile(c, 90):8.0f} cycles")
print("  positions per workgroup:", np.unique(raw[:, 7], return_counts=True))
print("  wall per workgroup median %.2f us" % (np.median(raw[:, 6] - raw[:, 0]) / 100))
for i, n in enumerate(["loader: vmcnt wait", "loader: barrier", "loader: issue", "loader: pieces/superstep"]):
    c = ld[:, i]
    print(f"  {n:24s} median {np.median(c):8.0f}  p10 {np.percentile(c, 10):8.0f}  p90 {np.percentile(c, 90):8.0f}")
