"""Phase stamps of k_fc_step_par (IDQN_FC_PROF=1, debug build (__graft_entry__.build_debug())): cycles of thread 0 between phase boundaries, per head."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IDQN_HIP_LIB", os.path.join(root, "i-dqn_amd", "libidqn_hip_debug.so"))
os.environ["IDQN_FC_PROF"] = "1"
sys.path[:0] = [root, os.path.join(root, "i-dqn_amd")]
import numpy as np, torch
from collections import namedtuple
from slimdqn.networks.idqn import iDQN
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
K, feats = 3, [100, 100]
agent = iDQN(0, 8, 4, K, feats, "fc", 3e-4, 0.99, 1, 1, 10**9, 10**9)
rng = np.random.default_rng(0)
b = Batch(torch.from_numpy(rng.standard_normal((32, 8)).astype(np.float32)).cuda(), torch.from_numpy(rng.integers(0, 4, 32).astype(np.int32)).cuda(),
          torch.from_numpy(rng.standard_normal(32).astype(np.float32)).cuda(), torch.from_numpy(rng.standard_normal((32, 8)).astype(np.float32)).cuda(),
          torch.from_numpy((rng.random(32) < 0.05).astype(np.uint8)).cuda())
for _ in range(30): agent._learn(b)
torch.cuda.synchronize()
st = agent._debug("fc_ws").cpu().numpy().view(np.int64)[: K * 16].reshape(K, 16)
names = ["zero LDS", "inputs + W to LDS", "fwd L0", "fwd L1", "fwd L2", None, None, "max, q, TD", "bwd L2", "bwd L1", "bwd L0", None, None, None, "(tail)"]
for k in range(K):
    d = np.diff(st[k])
    print(f"head {k}: total {st[k, 15] - st[k, 0]} cycles: " + "  ".join(f"{n} {int(x)}" for n, x in zip(names, d) if n and 0 < x < 10**7))
