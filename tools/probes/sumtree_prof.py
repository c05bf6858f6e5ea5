"""Phase stamps of k_sumtree_set (library built with -DST_PROF): wall-clock ticks (100 MHz) between its phases."""
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "i-dqn_amd")]
import numpy as np, torch
from slimdqn import _hip
lib, q = _hip.lib(), _hip.current_stream()
rng = np.random.default_rng(0)
depth = 21
nodes = torch.zeros(2**depth - 1, dtype=torch.float64, device="cuda")
scratch = torch.zeros(16 * 4096, dtype=torch.uint8, device="cuda")
for B in (32, 256, 1024):
    idx = torch.from_numpy(rng.integers(0, 1 << 20, B).astype(np.int32)).cuda()
    val = torch.from_numpy(rng.random(B)).cuda()
    acc = np.zeros(4)
    for rep in range(20):
        _hip.check(lib.sumtree_set(_hip.ptr(nodes), depth, _hip.ptr(idx), _hip.ptr(val), B, _hip.ptr(scratch), q), "set")
        torch.cuda.synchronize()
        st = scratch.view(torch.int64)[6000:6005].cpu().numpy()
        if rep >= 5:
            acc += np.diff(st) / 100.0
    print(B, "us: load+delta %.2f  sort %.2f  sorted deltas %.2f  runs %.2f" % tuple(acc / 15))
