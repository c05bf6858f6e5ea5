"""Compute side of the factored data-parallel step on ONE GPU, without communication: the step is issued exactly as
slimdqn/networks/parallel.py does, but the "gathered" factor buffer holds N copies of this rank's own factors, so the
fused Dense_0 update runs over N sample blocks like on rank 0 of an N-rank job.  Shows what the growing global batch
costs the update kernel (its MFMA work grows with N, its HBM traffic does not)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
from collections import namedtuple
import torch
import bench
from slimdqn import _hip
from slimdqn.networks.idqn import iDQN

Batch = namedtuple("Batch", "state action reward next_state is_terminal")
agent = iDQN(0, bench.OBS, bench.N_ACTIONS, bench.K_HEADS, bench.FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
b = Batch(*(torch.from_numpy(x).cuda() for x in bench.synthetic(1)))
lib, q = _hip.lib(), _hip.current_stream
K = agent._K
F, J = next(shape for name, _, shape in agent._leaves if name == "Dense_0/kernel")
X, Y = F * 32, J * 32
n_a3, n_dh = K * X, K * Y
for N in (1, 2, 4, 8):
    send = torch.zeros(n_a3 + n_dh, dtype=torch.float32, device="cuda")
    gathered = torch.zeros(N * (n_a3 + n_dh), dtype=torch.float32, device="cuda")
    def step():
        agent._learn(b, flags=_hip.F_STOP_BEFORE_DENSE0_WGRAD, mean_divisor=32 * N)
        _hip.check(lib.idqn_export_dense0_factors(agent._handle, _hip.ptr(send), _hip.ptr(send[n_a3:]), q()), "export")
        gathered[: n_a3 + n_dh].copy_(send)  # (stands for the collective; the other N - 1 slots keep their zeros / old values)
        _hip.check(lib.idqn_backward_rest(agent._handle, q()), "rest")
        args = (agent._handle, _hip.ptr(gathered), _hip.ptr(gathered[n_a3:]), N, 1, n_a3 + n_dh, X, X, n_a3 + n_dh, Y, Y)
        _hip.check(lib.idqn_finish_step_factored(*args, _hip.FACTORED_DENSE0, q()), "finish")
        _hip.check(lib.idqn_finish_step_factored(*args, _hip.FACTORED_REST, q()), "finish")
    for _ in range(20): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): step()
    e1.record(); torch.cuda.synchronize()
    print(f"N = {N}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per step (compute side of the factored step, no collective)", flush=True)
