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


def _worker(rank, world, port, mode, out, name="cnn_atari_a18_b64"):
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

        agent, bs, rec, _ = _agent(name)
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


@pytest.mark.parametrize("mode,name", [("factored", "cnn_atari_a18_b64"), ("overlap", "cnn_atari_a18_b64"),
                                       ("plain", "cnn_atari_a18_b64"), ("plain", "fc_lunar_k3")])
def test_two_rank_step_equals_the_single_device_step(tmp_path, mode, name):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path), name), nprocs=2, join=True)
    rec = json.load(open(os.path.join(GOLDEN, f"fp_path_{name}.json")))
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


def _hp_worker(rank, world, port, out):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "i-dqn_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from slimdqn.networks.head_parallel import HeadShardedIDQN

        agent = HeadShardedIDQN(9, *HP_ARGS)
        logs = _hp_run(agent)
        torch.cuda.synchronize()
        np.savez(f"{out}/hp{rank}.npz", online=agent._online.cpu().numpy(), target=agent._target.cpu().numpy(),
                 logs=np.asarray(logs, np.float64))
    finally:
        dist.destroy_process_group()


HP_OBS, HP_A, HP_K = (20, 20, 4), 5, 4
HP_ARGS = (HP_OBS, HP_A, HP_K, [32, 32, 32, 128], "cnn", 1e-3, 0.99, 1, 1, 4, 2)  # T = 4, D = 2


def _hp_run(agent):
    """9 environment steps of update_online / update_target on fixed batches; returns the logged losses."""
    from collections import namedtuple

    from oracle import qnet_ref as Q

    Batch = namedtuple("Batch", "state action reward next_state is_terminal")

    class Fixed:
        i = 0

        def sample(self):
            Fixed.i += 1
            return Batch(*Q.synthetic_batch(300 + Fixed.i, 32, HP_OBS, HP_A, "cnn"))

    Fixed.i = 0
    rb, logs = Fixed(), []
    for step in range(1, 10):
        agent.update_online_params(step, rb)
        has, lg = agent.update_target_params(step)
        if has:
            logs.append([lg["loss"]] + [lg[f"networks/{k}_loss"] for k in range(HP_K)])
    return logs


def test_two_rank_head_parallel_chain_equals_the_single_device_agent(tmp_path):
    """Head-parallel i-DQN on two ranks (2 heads each) through T-steps (copy + shift) and D-steps (sync) with the rows
    really travelling between processes: concatenated, the windows must equal the 4-head single-device agent."""
    import torch
    import torch.multiprocessing as mp

    from slimdqn.networks.idqn import iDQN

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_hp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref = iDQN(9, *HP_ARGS)
    ref_logs = _hp_run(ref)
    torch.cuda.synchronize()
    got = [np.load(f"{tmp_path}/hp{r}.npz") for r in range(2)]
    online = np.concatenate([g["online"] for g in got])
    target = np.concatenate([g["target"] for g in got])
    # window agents re-associate the split-K sums (fewer local heads): fp32 accumulation accuracy, not bits
    for name, a, b in (("online", online, ref._online.cpu().numpy()), ("target", target, ref._target.cpu().numpy())):
        err = np.abs(a - b)
        per_head = [(float((e <= 1e-6).mean()), float(e.max())) for e in err]
        assert (err <= 1e-6).mean() >= 0.999 and err.max() <= 2 * 1e-3 * 9, (name, per_head)
    np.testing.assert_allclose(got[0]["logs"], np.asarray(ref_logs), rtol=1e-4)
    np.testing.assert_array_equal(got[0]["logs"], got[1]["logs"])  # every rank logs all K heads
