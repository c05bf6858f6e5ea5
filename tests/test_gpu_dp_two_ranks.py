"""GPU, TWO ranks of the real data-parallel step on one card.

RCCL refuses two ranks on one GPU, so the transport here is gloo (which stages device tensors through the host); what
is under test is everything else of ``data_parallel_step`` on the device path: each rank runs the split backward on
its 32-sample shard of a 64-sample batch, the Dense_0 factors are really exchanged between two processes, the fused
update runs over the gathered blocks -- and both ranks must land on the single-device 64-sample golden step.
Both variants (factored exchange, gradient all-reduce) are covered.
"""
import json
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _worker(rank, world, port, mode, out):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "i-dqn_amd"), os.path.dirname(os.path.abspath(__file__))):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from slimdqn.networks.parallel import data_parallel_step, shard_of
        from test_gpu_fp_path import _agent

        agent, bs, rec, _ = _agent("cnn_atari_a18_b64")
        losses = None
        for batch in bs:
            kw = dict(mode="factored") if mode == "factored" else dict(mode="allreduce", overlap=(mode == "overlap"))
            losses = data_parallel_step(agent, shard_of(batch, rank, world), len(batch.action), **kw)
        torch.cuda.synchronize()
        flat = agent._flat(agent._online)
        np.savez(f"{out}/rank{rank}.npz", losses=losses.cpu().numpy(), count=agent._count.cpu().numpy(),
                 **{k.replace("/", "__"): v for k, v in flat.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["factored", "overlap", "plain"])
def test_two_rank_step_equals_the_single_device_step(tmp_path, mode):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    rec = json.load(open(os.path.join(GOLDEN, "fp_path_cnn_atari_a18_b64.json")))
    n_steps = len(rec["steps"])
    last = rec["steps"][n_steps - 1]
    got = [np.load(f"{tmp_path}/rank{r}.npz") for r in range(2)]
    for r in range(2):
        assert np.abs(got[r]["losses"] - np.asarray(last["losses"])).max() <= 1e-5, (mode, r)
        assert got[r]["count"].tolist() == [n_steps] * len(last["losses"])
        K = len(last["losses"])
        for leaf, d in last["leaves"].items():
            v = got[r][leaf.replace("/", "__")].reshape(K, -1)[:, d["idx"]]
            err = np.abs(v - np.asarray(d["param"]))
            assert (err <= 3e-7).mean() >= 0.98 and err.max() <= 2 * rec["hyper"]["lr"] * n_steps, (mode, r, leaf)
    # the replicas stay bit-identical: same blocks, same order, same arithmetic on both ranks
    for k in got[0].files:
        np.testing.assert_array_equal(got[0][k], got[1][k], err_msg=f"{mode}: ranks diverged in {k}")
