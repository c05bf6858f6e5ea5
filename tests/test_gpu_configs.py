"""BASELINE configs 4 and 5 at their full single-device workload, long-horizon drift, and RCCL stream ordering.

fp goldens are NOT reference-captured (oracle/__init__.py: parity unpinned): they come from the fp64 restatement.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOSS_ATOL = 1e-5  # north_star: fp32 loss within 1e-5


def _golden_agent(name):
    from test_gpu_fp_path import _agent

    return _agent(name)


from test_gpu_fp_path import conv_mode  # noqa: E402,F401  (fixture: both conv arithmetic modes)


def _oracle_preacts(ph, state):
    """fp64 pre-activations of the three convs of one head (architectures/dqn.py:43-51)."""
    from oracle import qnet_ref as Q

    a, ys = state.astype(np.float64) / 255.0, []
    for li, (k, s) in enumerate(Q.CNN_GEOM):
        y, _ = Q.conv_fwd(a, ph[f"Conv_{li}/kernel"].astype(np.float64), ph[f"Conv_{li}/bias"].astype(np.float64), s)
        ys.append(y)
        a = np.maximum(y, 0)
    return ys


@pytest.mark.parametrize("name", ["cnn_atari_k5_b256", "cnn_atari_k64"])
def test_full_size_configs_against_goldens(name, conv_mode):
    """Config 4's global batch (K = 5, B = 256: eight 32-sample blocks through every kernel) and config 5's head count
    (K = 64, B = 32) on ONE device, in both conv arithmetic modes: per-head losses within 1e-5, post-Adam parameters at the
    probe indices at fp32 accuracy -- EXCEPT where a ReLU unit's pre-activation lies within fp32 round-off of zero and the
    fp32 path takes the other branch than the fp64 oracle.  That exception is checked, not assumed: every head with a
    parameter probe off by more than 3e-7 must show at least one such flipped unit, and every flipped unit's fp64
    pre-activation must be no larger than fp32 round-off of its layer (1e-5 of the layer's largest pre-activation)."""
    from oracle import qnet_ref as Q
    from test_gpu_fp_path import _unpack_act, _unpack_planes

    agent, bs, rec, (arch, obs, A, feats, K, B, steps, p, pt, batches) = _golden_agent(name)
    losses = agent._learn(bs[0]).cpu().numpy()
    want = np.asarray(rec["steps"][0]["losses"])
    assert np.abs(losses - want).max() <= LOSS_ATOL, np.abs(losses - want).max()
    flat = agent._flat(agent._online)
    off_heads = set()
    for leaf, d in rec["steps"][0]["leaves"].items():
        err = np.abs(flat[leaf].reshape(K, -1)[:, d["idx"]] - np.asarray(d["param"]))
        off_heads |= set(np.nonzero((err > 3e-7).any(axis=1))[0].tolist())
        assert (err <= 3e-7).mean() >= 0.98 and err.max() <= 2 * rec["hyper"]["lr"], (leaf, err.max(), (err <= 3e-7).mean())
    # --- the ReLU-flip account ---
    planes = conv_mode == "bf16x3"
    nb = (B + 31) // 32
    H, W, C = obs
    geo = []
    for (k, s), f in zip(Q.CNN_GEOM, feats[:3]):
        oh, lh, hh = Q.same_pad(H, k, s)
        ow, lw, hw = Q.same_pad(W, k, s)
        geo.append(dict(IH=H, IW=W, CI=C, OH=oh, OW=ow, CO=f, lo_h=lh, hi_h=hh, lo_w=lw, hi_w=hw))
        H, W, C = oh, ow, f
    bufs = {}
    for li, nm in enumerate(["a1", "a2"]):  # online nets = the first K * nb slots of every activation buffer
        gi = geo[li + 1]
        Hp, Wp = gi["IH"] + gi["lo_h"] + gi["hi_h"], gi["IW"] + gi["lo_w"] + gi["hi_w"]
        per_slot = Hp * Wp * gi["CI"] * 32 * (3 if planes else 1) // (2 if planes else 1)  # float32 words per slot
        bufs[nm] = (agent._debug(nm + ("p" if planes else ""))[: K * nb * per_slot].cpu(), gi, Hp, Wp)
    go = geo[2]
    a3 = agent._debug("a3")[: K * nb * go["OH"] * go["OW"] * go["CO"] * 32].cpu()
    state = batches[0][0]
    checked = sorted(off_heads | {0, K - 1})
    flips_of = {}
    for k in checked:
        ys = _oracle_preacts(Q.head(p, k), state)
        n_flip = 0
        for li in range(3):
            got = []
            for bb in range(nb):
                if li < 2:
                    t, gi, Hp, Wp = bufs[["a1", "a2"][li]]
                    un = _unpack_planes if planes else _unpack_act
                    got.append(un(t, K * nb, k * nb + bb, gi["IH"], gi["IW"], gi["CI"], gi["lo_h"], gi["lo_w"], Hp, Wp))
                else:
                    got.append(_unpack_act(a3, K * nb, k * nb + bb, go["OH"], go["OW"], go["CO"], 0, 0, go["OH"], go["OW"]))
            act = np.concatenate(got, axis=0)[:B]
            y = ys[li]
            flipped = (act > 0) != (y > 0)
            n_flip += int(flipped.sum())
            if flipped.any():
                assert np.abs(y[flipped]).max() <= 1e-5 * np.abs(y).max(), (name, k, li, np.abs(y[flipped]).max(), np.abs(y).max())
        flips_of[k] = n_flip
    print(f"[{name} {conv_mode}] heads with a probe off by > 3e-7: {sorted(off_heads)}; flipped ReLU units per checked head: {flips_of}")
    for k in off_heads:
        assert flips_of[k] >= 1, (name, k, "parameter probes differ by more than 3e-7 although no ReLU unit took the other branch")


def test_config4_prioritized_learner_on_a_million_leaf_tree():
    """Config 4's sampling side at full size: a 2^20-leaf sum tree in HBM, B = 256 prioritized samples per step, TD-error
    write-back -- tree invariants after 20 steps on Atari-shaped frames (extension: no reference behaviour to match)."""
    import torch

    from slimdqn.networks.idqn import iDQN
    from slimdqn.sample_collection.per import PrioritizedLearner, SlotPrioritizedSampler
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement

    cap, B, K = 1 << 20, 256, 5
    sampler = SlotPrioritizedSampler(0, cap, priority_exponent=0.6)
    rb = ReplayBuffer(sampler, batch_size=B, max_capacity=cap, stack_size=4, update_horizon=1, gamma=0.99)
    agent = iDQN(0, (84, 84, 4), 6, K, [32, 64, 64, 512], "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
    rng = np.random.default_rng(0)
    n_add = 3000
    for i in range(n_add):
        rb.add(TransitionElement(rng.integers(0, 256, (84, 84), dtype=np.uint8), int(rng.integers(6)),
                                 float(rng.integers(-1, 2)), bool(i % 400 == 399), False))
    learner = PrioritizedLearner(agent, rb, beta=0.4, eps=1e-3, reduce="mean")
    for _ in range(20):
        losses = learner.step()
    torch.cuda.synchronize()
    assert np.isfinite(losses.cpu().numpy()).all()
    tree = sampler._sum_tree
    nodes = tree._nodes
    leaves = nodes[tree._first_leaf_offset : tree._first_leaf_offset + cap]
    assert (leaves[len(sampler):] == 0).all() and (leaves[: len(sampler)] > 0).all()
    assert abs(nodes[0] - leaves.sum()) <= 1e-9 * leaves.sum()  # the root is the sum of the leaves (21 levels)
    lv = learner._leaves.cpu().numpy()
    assert lv.min() >= 0 and lv.max() < len(sampler)


def test_hundred_step_drift_against_the_oracle():
    """100 consecutive fused steps on cnn_small (hardware rcp / sqrt in Adam, bf16x3 convs, fused Dense_0 data gradient),
    free-running, against the oracle from the same start.  Two references: the oracle in fp32 (numpy float32 arithmetic,
    what the reference's XLA-CPU executable does) must be tracked within 1e-5 at EVERY step; the fp64 oracle within 1e-5
    over the first 50 steps -- beyond that ANY fp32 implementation, the numpy one included, has drifted from the fp64
    trajectory by up to 1e-4 (parameters are rounded to fp32 after every update; tools/probes/drift.py prints all three)."""
    from collections import namedtuple

    from oracle import make_golden as G
    from oracle import qnet_ref as Q
    from slimdqn.networks.idqn import iDQN

    name = "cnn_small"
    arch, obs, A, feats, K, B, _ = G.FP_CASES[name]
    p, pt, _ = G.fp_case_inputs(name)
    h = G.FP_HYPER
    agent = iDQN(0, obs, A, K, feats, arch, h["lr"], h["gamma"], h["n"], 1, 10**9, 10**9, adam_eps=h["eps"])
    agent._load_flat(agent._online, p)
    agent._load_flat(agent._target, pt)
    Batch = namedtuple("Batch", "state action reward next_state is_terminal")

    def fresh(dt):
        P = {n: a.astype(dt) for n, a in p.items()}
        return (P, {n: np.zeros_like(a) for n, a in P.items()}, {n: np.zeros_like(a) for n, a in P.items()}, np.zeros(K, np.int64))

    s64, s32 = fresh(np.float64), fresh(np.float32)
    pt32 = {n: a.astype(np.float32) for n, a in pt.items()}
    g_n = h["gamma"] ** h["n"]
    e32, e64 = [], []
    for s in range(100):
        batch = Q.synthetic_batch(1000 + s, B, obs, A, arch)
        got = agent._learn(Batch(*batch)).cpu().numpy()
        out = Q.learn_on_batch(s64[0], pt, s64[1], s64[2], s64[3], batch, arch, g_n, h["lr"], h["eps"], np.float64)
        s64, want64 = out[:4], out[4]
        out = Q.learn_on_batch(s32[0], pt32, s32[1], s32[2], s32[3], batch, arch, g_n, h["lr"], h["eps"], np.float32)
        s32, want32 = out[:4], out[4]
        e32.append(float(np.abs(got - want32).max()))
        e64.append(float(np.abs(got - want64).max()))
        assert e32[-1] <= LOSS_ATOL, (s, got, want32)
        assert s >= 50 or e64[-1] <= LOSS_ATOL, (s, got, want64)
    flat = agent._flat(agent._online)
    drift = max(float(np.abs(flat[n] - s32[0][n]).max()) for n in flat)
    print(f"\n100 steps: worst loss error vs fp32 oracle {max(e32):.2e}, vs fp64 oracle {max(e64):.2e} (first 50: {max(e64[:50]):.2e}); "
          f"largest parameter difference to the fp32 oracle {drift:.2e}")
    assert drift <= 2 * h["lr"]  # a ReLU that flips near zero moves a weight by a fraction of one update


@pytest.mark.parametrize("mode", ["native-side", "native-inline", "factored", "allreduce"])
def test_rccl_stream_ordering_stress(mode):
    """World-size-1 RCCL, 200 steps of B = 128 (four blocks per rank) per mode: the asynchronous step (collectives on
    RCCL's stream under the conv backward / the fused update) must equal, BIT FOR BIT, the same step with every
    collective waited for and the device synchronised around it.  A missing stream dependency shows as a difference.
    native-*: the whole schedule issued by ONE C call (idqn_dp_step: the library's own RCCL communicator, collectives on its
    side stream / on the compute stream) against the serialised Python schedule over torch.distributed -- its oracle."""
    import socket

    import torch
    import torch.distributed as dist
    from collections import namedtuple

    from oracle import qnet_ref as Q
    from slimdqn.networks.idqn import iDQN
    from slimdqn.networks.parallel import data_parallel_step

    if not dist.is_initialized():
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        owns = True
    else:
        owns = False
    try:
        arch, obs, A, feats, K, B = "cnn", (84, 84, 4), 6, [32, 64, 64, 512], 2, 128
        Batch = namedtuple("Batch", "state action reward next_state is_terminal")
        batches = [Batch(*(torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in Q.synthetic_batch(50 + i, B, obs, A, arch)))
                   for i in range(4)]
        agents = [iDQN(0, obs, A, K, feats, arch, 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4) for _ in range(2)]
        for s in range(200):
            for agent, serial in zip(agents, (False, True)):
                if mode.startswith("native"):
                    data_parallel_step(agent, batches[s % 4], B, mode="factored" if serial else "native", serial=serial,
                                       streams=mode.split("-")[1])
                else:
                    data_parallel_step(agent, batches[s % 4], B, mode=mode, serial=serial,
                                       overlap=None if mode == "factored" else True)
        torch.cuda.synchronize()
        a, b = agents[0]._online.cpu().numpy(), agents[1]._online.cpu().numpy()
        assert np.isfinite(a).all()
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), int((a.view(np.uint32) != b.view(np.uint32)).sum())
        assert (agents[0]._mu.cpu().numpy().view(np.uint32) == agents[1]._mu.cpu().numpy().view(np.uint32)).all()
        assert (agents[0]._cum.cpu().numpy() == agents[1]._cum.cpu().numpy()).all() and (agents[0]._count == agents[1]._count).all()
        for ag in agents:  # the library-side communicators go before the process group
            ag._destroy_handle()
    finally:
        if owns:
            dist.destroy_process_group()
