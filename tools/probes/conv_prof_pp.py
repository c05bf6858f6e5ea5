"""Phase stamps of the persistent plane conv kernel (convp_pp.hip) for one role (IDQN_CONV_PROF=role, debug build (__graft_entry__.build_debug())):
per workgroup, cycles of wave 0 and wave 4 by what they were doing, summed over the workgroup's items."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IDQN_HIP_LIB", os.path.join(root, "i-dqn_amd", "libidqn_hip_debug.so"))
sys.path[:0] = [root, os.path.join(root, "i-dqn_amd")]
import numpy as np
import torch
from collections import namedtuple

import bench
from slimdqn.networks.idqn import iDQN

role = int(os.environ["IDQN_CONV_PROF"])
K, B = int(os.environ.get("CPROF_HEADS", bench.K_HEADS)), int(os.environ.get("CPROF_BATCH", 256))
agent = iDQN(0, bench.OBS, bench.N_ACTIONS, K, bench.FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
b = Batch(*(torch.from_numpy(x).cuda() for x in bench.synthetic(1, batch=B)))
for _ in range(10):
    agent._learn(b)
torch.cuda.synchronize()
allp = agent._debug("cprof").cpu().numpy().view(np.int64).reshape(2, 4096, 8)
for plane, nm in ((0, "wave 0"), (1, "wave 4")):
    r0, r1 = allp[plane, :256], allp[plane, 1024:1280]
    ok = r0[:, 0] != 0
    r0, r1 = r0[ok], r1[ok]
    items = r0[:, 7]
    print(f"role {role} {nm}: {len(r0)} workgroups, items per workgroup {np.unique(items, return_counts=True)}, "
          f"launch span {(r0[:, 6].max() - r0[:, 0].min()) / 100:.1f} us, wall per workgroup median {np.median(r0[:, 6] - r0[:, 0]) / 100:.1f} us")
    per = lambda c: np.median(c / items)
    print(f"   per item (median, cycles): compute role {per(r0[:, 1]):7.0f} (of it waiting at barriers {per(r0[:, 2]):6.0f}) | loader role: "
          f"vmcnt wait {per(r0[:, 3]):6.0f}  barrier {per(r0[:, 4]):6.0f}  issue {per(r0[:, 5]):6.0f}  epilogue slices {per(r1[:, 0]):6.0f}  "
          f"decode {per(r1[:, 1]):6.0f} | tail epilogue {np.median(r1[:, 2]):6.0f} per workgroup")
