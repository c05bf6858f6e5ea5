"""Phase stamps of the TD / loss kernel (IDQN_CONV_PROF=9): cycles between its barriers, median over workgroups."""
import os
os.environ.setdefault("IDQN_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "i-dqn_amd", "libidqn_hip_debug.so"))  # debug build (__graft_entry__.build_debug()): the stamps / switches used here
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from collections import namedtuple
import bench
from slimdqn.networks.idqn import iDQN
A = int(sys.argv[1]) if len(sys.argv) > 1 else 6
agent = iDQN(0, bench.OBS, A, bench.K_HEADS, bench.FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
b = Batch(*(torch.from_numpy(x).cuda() for x in bench.synthetic(1, A)))
for _ in range(30): agent._learn(b)
torch.cuda.synchronize()
p = agent._debug("cprof").cpu().numpy().view(np.int64).reshape(-1, 8)[: 5 * 16]
d = np.diff(p[:, :6], axis=1)
print(f"A = {A}: median cycles  q-reduce {np.median(d[:,0]):.0f}  td (wave 0) {np.median(d[:,1]):.0f}  dh {np.median(d[:,2]):.0f}  gw {np.median(d[:,3]):.0f}  gb1 + stores {np.median(d[:,4]):.0f}  total {np.median(p[:,5]-p[:,0]):.0f}")
