"""Throughput of the integer / byte side of the path on the GPU ("sum-tree / replay gather ...
reported separately in GB/s"), and the sample -> gather -> learn pipeline rate.

    python tools/bench_sampling.py [--capacity 1048576] [--reps 200] > gpurun_out/sampling.json

Rows A11-A14: ReplayBuffer.sample's gather (replay_buffer.py:223-229), SumTree.set / query (sum_tree.py:20-102).
All inputs are resident in HBM; each op is timed with HIP events on the launch stream over `reps` back-to-back calls.
Algorithmic bytes: gather = 2 * B * obs_bytes read + the same written; query = B * (depth - 1) node reads of 8 B
(dependent: latency-bound, not bandwidth-bound); set = B * depth read-modify-writes of 8 B.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "i-dqn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402


def timed(fn, reps):
    import torch

    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps  # seconds per call (launches on torch's current stream)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--capacity", type=int, default=1 << 20)
    ap.add_argument("--slots", type=int, default=1 << 16, help="replay slots allocated for the gather test (56 KB each)")
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()

    import torch

    from slimdqn import _hip

    lib, q = _hip.lib(), _hip.current_stream()
    out = {"device": torch.cuda.get_device_name(0), "capacity": args.capacity}
    rng = np.random.default_rng(0)

    # ---- replay gather (A11) -------------------------------------------------------------------
    obs_bytes = 84 * 84 * 4
    store = torch.randint(0, 256, (args.slots, 2, obs_bytes), dtype=torch.uint8, device="cuda")
    for B in (32, 256, 2048):
        slots = torch.from_numpy(rng.integers(0, args.slots, B).astype(np.int32)).cuda()
        s_out = torch.empty((B, obs_bytes), dtype=torch.uint8, device="cuda")
        s2_out = torch.empty((B, obs_bytes), dtype=torch.uint8, device="cuda")

        def gather():
            _hip.check(lib.replay_gather(_hip.ptr(store), obs_bytes, _hip.ptr(slots), B, _hip.ptr(s_out),
                                         _hip.ptr(s2_out), q), "replay_gather")

        dt = timed(gather, args.reps)
        ref = store[slots.long()]
        assert torch.equal(ref[:, 0], s_out) and torch.equal(ref[:, 1], s2_out)
        moved = 2 * 2 * B * obs_bytes  # read + written
        out[f"replay_gather_B{B}"] = {"us": dt * 1e6, "GB/s": moved / dt / 1e9, "bytes": moved}

    # ---- sum tree (A14) ------------------------------------------------------------------------
    depth = int(np.ceil(np.log2(args.capacity))) + 1
    nodes = torch.zeros(2**depth - 1, dtype=torch.float64, device="cuda")
    scratch = torch.empty(16 * 4096, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    # fill the tree: 4096 leaves per set call
    pri = rng.random(args.capacity) + 0.01
    t0 = time.perf_counter()
    for lo in range(0, args.capacity, 4096):
        idx = torch.arange(lo, min(lo + 4096, args.capacity), dtype=torch.int32, device="cuda")
        val = torch.from_numpy(pri[lo : lo + 4096]).cuda()
        _hip.check(lib.sumtree_set(_hip.ptr(nodes), depth, _hip.ptr(idx), _hip.ptr(val), idx.numel(), _hip.ptr(scratch), q),
                   "sumtree_set")
    torch.cuda.synchronize()
    out["sumtree_fill_s"] = time.perf_counter() - t0
    root = float(nodes[0].item())
    assert abs(root - pri.sum()) <= 1e-6 * root, (root, pri.sum())
    for B in (32, 256, 4096):
        targets = torch.from_numpy(rng.uniform(0, root * (1 - 1e-9), B)).cuda()
        leaves = torch.empty(B, dtype=torch.int32, device="cuda")

        def query():
            _hip.check(lib.sumtree_query(_hip.ptr(nodes), depth, _hip.ptr(targets), B, _hip.ptr(leaves), _hip.ptr(status), q),
                       "sumtree_query")

        dt = timed(query, args.reps)
        assert int(status.item()) == 0
        out[f"sumtree_query_B{B}"] = {"us": dt * 1e6, "Mqueries/s": B / dt / 1e6,
                                      "GB/s": B * (depth - 1) * 8 / dt / 1e9, "depth": depth}
        idx = torch.from_numpy(rng.integers(0, args.capacity, B).astype(np.int32)).cuda()
        val = torch.from_numpy(rng.random(B)).cuda()

        def tset():
            _hip.check(lib.sumtree_set(_hip.ptr(nodes), depth, _hip.ptr(idx), _hip.ptr(val), B, _hip.ptr(scratch), q),
                       "sumtree_set")

        dt = timed(tset, args.reps)
        out[f"sumtree_set_B{B}"] = {"us": dt * 1e6, "Mupdates/s": B / dt / 1e6, "GB/s": B * depth * 16 / dt / 1e9}

    # ---- sample -> gather -> learn, everything device-side (prioritized, config-4-like on one GPU) ----
    from collections import namedtuple

    from slimdqn.networks.idqn import iDQN

    agent = iDQN(0, (84, 84, 4), 6, 5, [32, 64, 64, 512], "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
    Batch = namedtuple("Batch", "state action reward next_state is_terminal")
    B = 32
    act = torch.from_numpy(rng.integers(0, 6, args.slots).astype(np.int32)).cuda()
    rew = torch.from_numpy(rng.integers(-1, 2, args.slots).astype(np.float32)).cuda()
    ter = torch.from_numpy((rng.random(args.slots) < 0.01).astype(np.uint8)).cuda()
    s_out = torch.empty((B, 84, 84, 4), dtype=torch.uint8, device="cuda")
    s2_out = torch.empty((B, 84, 84, 4), dtype=torch.uint8, device="cuda")
    a_out = torch.empty(B, dtype=torch.int32, device="cuda")
    r_out = torch.empty(B, dtype=torch.float32, device="cuda")
    t_out = torch.empty(B, dtype=torch.uint8, device="cuda")
    leaves = torch.empty(B, dtype=torch.int32, device="cuda")
    slots_dev = torch.empty(B, dtype=torch.int32, device="cuda")
    host_rng = np.random.default_rng(1)
    targets_pinned = torch.empty(B, dtype=torch.float64).pin_memory()
    targets_dev = torch.empty(B, dtype=torch.float64, device="cuda")

    def pipeline():
        # host draws the B uniforms (samplers.py:110), the tree descent, the gather and the step stay on the device
        targets_pinned.copy_(torch.from_numpy(host_rng.uniform(0.0, root * (1 - 1e-9), B)))
        targets_dev.copy_(targets_pinned, non_blocking=True)
        _hip.check(lib.sumtree_query(_hip.ptr(nodes), depth, _hip.ptr(targets_dev), B, _hip.ptr(leaves), _hip.ptr(status), q),
                   "sumtree_query")
        torch.remainder(leaves, args.slots, out=slots_dev)  # leaf index -> slot of this (smaller) test store
        _hip.check(lib.replay_gather(_hip.ptr(store), obs_bytes, _hip.ptr(slots_dev), B, _hip.ptr(s_out), _hip.ptr(s2_out), q),
                   "replay_gather")
        _hip.check(lib.replay_gather_scalars(_hip.ptr(act), _hip.ptr(rew), _hip.ptr(ter), _hip.ptr(slots_dev), B,
                                             _hip.ptr(a_out), _hip.ptr(r_out), _hip.ptr(t_out), q), "replay_gather_scalars")
        agent._learn(Batch(s_out, a_out, r_out, s2_out, t_out))

    for _ in range(20):
        pipeline()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        pipeline()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out["sample_gather_learn"] = {"ms_per_step": dt * 1e3, "grad-steps/s": 1 / dt,
                                  "what": "PER query (cap 2^20) + gather + K=5 B=32 Nature-CNN step, wall clock"}
    # ---- ReplayBuffer.add / sample through the Python mirror (frame ring: one 7 KB upload per environment step) ----
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
    from slimdqn.sample_collection.samplers import UniformSamplingDistribution

    rb = ReplayBuffer(UniformSamplingDistribution(0), batch_size=32, max_capacity=100_000, stack_size=4,
                      update_horizon=1, gamma=0.99)
    frames = rng.integers(0, 256, size=(64, 84, 84), dtype=np.uint8)
    for i in range(2000):
        rb.add(TransitionElement(frames[i % 64], i % 6, 1.0, i % 500 == 499, False))
    torch.cuda.synchronize()
    n = 20000
    t0 = time.perf_counter()
    for i in range(n):
        rb.add(TransitionElement(frames[i % 64], i % 6, 1.0, i % 500 == 499, False))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out["replay_add"] = {"us_per_add": dt * 1e6, "adds/s": 1 / dt, "bytes_uploaded_per_add": 84 * 84,
                         "ring_frames": rb._n_frames, "store_MB": rb._frames.numel() / 1e6}
    t0 = time.perf_counter()
    for i in range(2000):
        rb.sample()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2000
    out["replay_sample_B32"] = {"us_per_sample": dt * 1e6, "what": "host PCG64 draw + key map + stacked gather launch"}

    def gather_only():
        _hip.check(lib.replay_gather_stacked(_hip.ptr(rb._frames), rb._n_frames, rb._frame_elems, 1, 4,
                                             _hip.ptr(rb._meta_dev), _hip.ptr(rb._stage[32]["slots"]), 32,
                                             _hip.ptr(rb._stage[32]["state"]), _hip.ptr(rb._stage[32]["next_state"]),
                                             _hip.ptr(rb._stage[32]["action"]), _hip.ptr(rb._stage[32]["reward"]),
                                             _hip.ptr(rb._stage[32]["terminal"]), q), "replay_gather_stacked")

    dt = timed(gather_only, args.reps)
    moved = 2 * 2 * 32 * 84 * 84 * 4
    out["replay_gather_stacked_B32"] = {"us": dt * 1e6, "GB/s": moved / dt / 1e9, "bytes": moved}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
