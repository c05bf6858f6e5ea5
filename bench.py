"""bench.py -- i-DQN gradient-steps/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (iDQN.learn_on_batch, reference slimdqn/networks/idqn.py:96-109:
2K forwards + K backwards + K Adam updates, per-head losses produced) over one synthetic minibatch
that is already resident in HBM (K=5, B=32, uint8 84x84x4, A=6, features [32,64,64,512]).

N = 1: the fused path (Dense_0 weight gradient + Adam in one kernel).
N > 1: data-parallel, weak scaling: every rank takes its own 32-sample shard of a global batch of 32*N.
       Default ("factored", slimdqn/networks/parallel.py): the Dense_0 gradient factors a3 / dh (5.3 MB per rank)
       are all-gathered over RCCL while the conv backward runs, the 1.6 MB of small-leaf gradients are
       all-reduced, and every rank runs the fused Dense_0 weight-gradient + Adam kernel over the gathered global
       batch.  IDQN_DP_MODE=allreduce all-reduces the 80.9 MB gradient arena instead.
       `value` counts 32-sample gradient steps: N per global step (units all ranks processed / time).

roofline: the dominant kernel (the fused Dense_0 update, HBM-bound) timed with hipEvents on its own stream inside
the timed regions, on every 4th step (PROFILE_EVERY: the bracket keeps the launch from overlapping its neighbours and
costs a bracketed step 4.5 - 5 us); cpu_baseline: the oracle's torch-CPU fp32 restatement of the same step, timed on this
box's host cores on a bounded sample (rank 0, N = 1 only) -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "i-dqn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

K_HEADS, BATCH, N_ACTIONS, OBS, FEATURES = 5, 32, 6, (84, 84, 4), [32, 64, 64, 512]
PROFILE_EVERY = 4  # every 4th step of a timed region carries the hipEvent bracket of the dominant kernel (roofline.achieved)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK = 2.5e15  # dense bf16 MFMA (MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3e12  # f32-input MFMA / f32 vector rate; the step's FLOPs are f32 FLOPs whatever the matrix cores run


def synthetic(seed, n_actions=N_ACTIONS, batch=BATCH):
    """SURVEY 8d inputs: iid uint8 frames, uniform actions, rewards in {-1,0,1}, 1 % terminals."""
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 256, size=(batch,) + OBS, dtype=np.uint8)
    s2 = rng.integers(0, 256, size=(batch,) + OBS, dtype=np.uint8)
    a = rng.integers(0, n_actions, size=batch).astype(np.int32)
    r = rng.integers(-1, 2, size=batch).astype(np.float32)
    t = (rng.random(batch) < 0.01).astype(np.uint8)
    return s, a, r, s2, t


def csrc_digest():
    """sha256 over the kernel sources (i-dqn_amd/csrc/*, include/*.h: names and contents, sorted): what a PMC summary is valid for."""
    import glob
    import hashlib

    hsh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "i-dqn_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        hsh.update(os.path.basename(f).encode())
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()


def step_work(K, B, A):
    """Whole-step algorithmic work: FLOPs = K B (2 F_fwd + F_bwd), bytes = K 7 4 P + 2 B 28224 + 9 B."""
    macs = 3612672 + 3964928 + 4460544 + 3964928 + 512 * A
    f_fwd = 2 * macs
    f_bwd = 2 * f_fwd - 2 * 3612672
    P = 8 * 8 * 4 * 32 + 32 + 4 * 4 * 32 * 64 + 64 + 3 * 3 * 64 * 64 + 64 + 7744 * 512 + 512 + 512 * A + A
    return K * B * (2 * f_fwd + f_bwd), K * 7 * 4 * P + 2 * B * 28224 + 9 * B


def heads_fit(A, heads=(1, 2, 3, 5), steps=200, warmup=30):
    """The K-independent part of the step as a tracked number: the plain step at K = 1, 2, 3, 5 heads (K = 1 is DQN,
    slimdqn/networks/dqn.py:60-73), B = 32, least-squares line us(K) = fixed_us + per_head_us * K."""
    import torch

    from collections import namedtuple

    from slimdqn.networks.idqn import iDQN

    Batch = namedtuple("Batch", "state action reward next_state is_terminal")
    batches = [Batch(*(torch.from_numpy(x).cuda() for x in synthetic(2000 + i, A))) for i in range(4)]
    us = []
    for K in heads:
        agent = iDQN(0, OBS, A, K, FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
        for i in range(warmup):
            agent._learn(batches[i % 4])
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                agent._learn(batches[i % 4])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps * 1e6
            best = dt if best is None else min(best, dt)
        us.append(best)
        agent._destroy_handle()
        del agent
    slope, icpt = np.polyfit(np.asarray(heads, np.float64), np.asarray(us), 1)
    flops1, bytes1 = step_work(1, BATCH, A)
    return {"heads": list(heads), "us_per_step": us, "fixed_us": float(icpt), "per_head_us": float(slope),
            "k1_floor_us_mfma_f32": flops1 / MFMA_F32_PEAK * 1e6, "k1_floor_us_hbm": bytes1 / (HBM_PEAK_GBS * 1e9) * 1e6,
            "what": f"plain step, B = 32, best of 3 x {steps} steps per K; line fit over K = {list(heads)}"}


def usable_cores():
    """Host cores this process may really use: affinity, capped by the cgroup CPU quota and by the
    16-core share a one-GPU box gives (an over-subscribed torch pool is far slower than a right-sized one)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("IDQN_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(budget_s=15.0, n_actions=N_ACTIONS, heads=K_HEADS, batch=BATCH):
    """The oracle's torch-CPU fp32 restatement (K heads batched), all host cores, bounded sample."""
    import torch

    from oracle import qnet_ref as Q
    from oracle import torch_ref as T

    cores = usable_cores()
    torch.set_num_threads(cores)
    p = Q.init_params(0, "cnn", OBS, n_actions, FEATURES, heads)
    pt = Q.init_params(1, "cnn", OBS, n_actions, FEATURES, heads)
    step = T.BatchedStep(p, pt, n_actions, 0.99, 6.25e-5, 1.5e-4)
    s, a, r, s2, t = synthetic(0, n_actions, batch)
    b = (s, a, r, s2, t.astype(bool))
    step.step(b)
    n, t0 = 0, time.perf_counter()
    while True:
        step.step(b)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 200:
            break
    return {"value": n / dt, "unit": "grad-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of the same K={heads} B={batch} Nature-CNN step in {dt:.1f} s (oracle/torch_ref.BatchedStep, "
                      f"torch-CPU fp32, {cores} threads, timed before the GPU regions; the JAX leg is probed separately: cpu_baseline.jax)"}


def sampling_leg(reps=300):
    """the integer / byte side of the path, reported separately (rank 0, N = 1): the stacked replay gather of a
    32-sample minibatch out of the HBM frame ring (replay_buffer.py:223-229), SumTree.query and SumTree.set on a
    2^20-leaf tree (sum_tree.py:20-102).  HIP events on the launch stream over `reps` back-to-back calls each."""
    import torch

    from slimdqn import _hip

    lib, q = _hip.lib(), _hip.current_stream()
    rng = np.random.default_rng(0)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    out = {}
    n_frames, fe, stack, cap = 1 << 15, 84 * 84, 4, 1 << 14
    frames = torch.randint(0, 256, (n_frames, fe), dtype=torch.uint8, device="cuda")
    meta = np.zeros((cap, 8), np.int32)
    meta[:, 0] = rng.integers(8, n_frames - 8, cap)
    meta[:, 1] = 4
    meta[:, 2] = meta[:, 0] + 1
    meta[:, 3] = 4
    meta_dev = torch.from_numpy(meta).cuda()
    for B in (32, 256):
        slots = torch.from_numpy(rng.integers(0, cap, B).astype(np.int32)).cuda()
        so = torch.empty((B, fe, stack), dtype=torch.uint8, device="cuda")
        s2o = torch.empty((B, fe, stack), dtype=torch.uint8, device="cuda")
        ao = torch.empty(B, dtype=torch.int32, device="cuda")
        ro = torch.empty(B, dtype=torch.float32, device="cuda")
        to = torch.empty(B, dtype=torch.uint8, device="cuda")

        def gather():
            _hip.check(lib.replay_gather_stacked(_hip.ptr(frames), n_frames, fe, 1, stack, _hip.ptr(meta_dev), _hip.ptr(slots), B,
                                                 _hip.ptr(so), _hip.ptr(s2o), _hip.ptr(ao), _hip.ptr(ro), _hip.ptr(to), q),
                       "replay_gather_stacked")

        dt = timed(gather)
        moved = 2 * 2 * B * fe * stack  # frames read + stacks written, state and next_state
        out[f"gather_B{B}"] = {"us": dt * 1e6, "GBps": moved / dt / 1e9, "bytes": moved}
    depth = 21
    nodes = torch.zeros(2**depth - 1, dtype=torch.float64, device="cuda")
    scratch = torch.empty(16 * 4096, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    pri = rng.random(1 << 20) + 0.01
    for lo in range(0, 1 << 20, 4096):
        idx = torch.arange(lo, lo + 4096, dtype=torch.int32, device="cuda")
        val = torch.from_numpy(pri[lo : lo + 4096]).cuda()
        _hip.check(lib.sumtree_set(_hip.ptr(nodes), depth, _hip.ptr(idx), _hip.ptr(val), 4096, _hip.ptr(scratch), q), "sumtree_set")
    root = float(nodes[0].item())
    for B in (32, 256):
        targets = torch.from_numpy(rng.uniform(0, root * (1 - 1e-9), B)).cuda()
        leaves = torch.empty(B, dtype=torch.int32, device="cuda")
        idx = torch.from_numpy(rng.integers(0, 1 << 20, B).astype(np.int32)).cuda()
        val = torch.from_numpy(rng.random(B)).cuda()
        dq = timed(lambda: _hip.check(lib.sumtree_query(_hip.ptr(nodes), depth, _hip.ptr(targets), B, _hip.ptr(leaves),
                                                        _hip.ptr(status), q), "sumtree_query"))
        ds = timed(lambda: _hip.check(lib.sumtree_set(_hip.ptr(nodes), depth, _hip.ptr(idx), _hip.ptr(val), B, _hip.ptr(scratch), q),
                                      "sumtree_set"))
        out[f"sumtree_B{B}"] = {"query_us": dq * 1e6, "set_us": ds * 1e6, "leaves": 1 << 20, "depth": depth,
                                "query_GBps": B * (depth - 1) * 8 / dq / 1e9, "set_GBps": B * depth * 16 / ds / 1e9}
    # reference protocol on the product class (samplers.py:52-116): host wall clock per call, results on the host
    import time

    from slimdqn.sample_collection.samplers import PrioritizedSamplingDistribution

    ps = PrioritizedSamplingDistribution(0, 1 << 20, 1.0)
    for k in range(4096):
        ps.add(k, priority=float(pri[k]))
    torch.cuda.synchronize()

    def wall(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    nk = [4096]

    def add_remove():
        ps.remove(nk[0] - 4096)
        ps.add(nk[0], priority=0.5)
        nk[0] += 1

    out["prioritized_protocol"] = {"sample_B32_us": wall(lambda: ps.sample(32), 200), "sample_B256_us": wall(lambda: ps.sample(256), 200),
                                   "remove_plus_add_us": wall(add_remove, 200), "leaves": 1 << 20,
                                   "what": "PrioritizedSamplingDistribution on a 2^20-leaf HBM tree, host wall clock per call, keys "
                                           "returned to the host: sample = one launch + one polled mailbox read; remove + add = two "
                                           "launches, no host read"}
    out["what"] = ("replay_gather_stacked out of a 2^15-frame ring (bytes = frames read + stacks written); sum tree of 2^20 "
                   "leaves: query = B x 20 dependent 8-byte reads, set = B x 21 read-modify-writes; latency-bound, bytes for scale")
    return out


def jax_cpu_probe(budget_s=10.0, n_actions=N_ACTIONS):
    """BASELINE.md 2.2 names a JAX-CPU leg.  jax is not installed in the build container; this probes for it on the box the
    bench runs on (JAX_PLATFORMS=cpu) and, if it imports, times a from-scratch jit(vmap(value_and_grad) + Adam) of the same
    step on the same synthetic inputs (slimdqn/networks/idqn.py:96-109 is the shape of it: no reference file is used) and
    returns per-head losses + a few gradient probes of the small golden case -- vectors this build did not author.
    Otherwise the reason is recorded ("ImportError: ...")."""
    os.environ.setdefault("JAX_PLATFORMS", "cpu")
    try:
        import jax
        import jax.numpy as jnp
    except Exception as e:  # noqa: BLE001
        return {"status": f"{type(e).__name__}: {e}"}
    try:
        from oracle import make_golden as G
        from oracle import qnet_ref as Q

        def net(p, x):
            a = x.astype(jnp.float32) / 255.0
            for li, (k, s) in enumerate(Q.CNN_GEOM):
                a = jax.lax.conv_general_dilated(a, p[f"Conv_{li}/kernel"], (s, s), "SAME", dimension_numbers=("NHWC", "HWIO", "NHWC"))
                a = jax.nn.relu(a + p[f"Conv_{li}/bias"])
            a = a.reshape(a.shape[0], -1)
            a = jax.nn.relu(a @ p["Dense_0/kernel"] + p["Dense_0/bias"])
            return a @ p["Dense_1/kernel"] + p["Dense_1/bias"]

        def loss(p, pt, batch, gamma_n):
            s, a, r, s2, t = batch
            tgt = r + (1.0 - t) * gamma_n * net(pt, s2).max(1)
            q = jnp.take_along_axis(net(p, s), a[:, None], axis=1)[:, 0]
            return jnp.mean((q - tgt) ** 2)

        def step(p, pt, m, v, count, batch, lr, eps, gamma_n):
            def one(pk, ptk, mk, vk, ck):
                l, g = jax.value_and_grad(loss)(pk, ptk, batch, gamma_n)
                t = ck + 1
                mk = jax.tree_util.tree_map(lambda mm, gg: 0.1 * gg + 0.9 * mm, mk, g)
                vk = jax.tree_util.tree_map(lambda vv, gg: 0.001 * gg * gg + 0.999 * vv, vk, g)
                bc1, bc2 = 1 - 0.9 ** t, 1 - 0.999 ** t
                pk = jax.tree_util.tree_map(lambda th, mm, vv: th - lr * (mm / bc1) / (jnp.sqrt(vv / bc2) + eps), pk, mk, vk)
                return pk, mk, vk, t, l, g
            return jax.vmap(one)(p, pt, m, v, count)

        jstep = jax.jit(step, static_argnums=(6, 7, 8))

        def run(p, pt, batch):
            pj = {n: jnp.asarray(a) for n, a in p.items()}
            ptj = {n: jnp.asarray(a) for n, a in pt.items()}
            z = {n: jnp.zeros_like(a) for n, a in pj.items()}
            K = next(iter(p.values())).shape[0]
            s, a, r, s2, t = batch
            b = (jnp.asarray(s), jnp.asarray(a.astype(np.int32)), jnp.asarray(r), jnp.asarray(s2), jnp.asarray(t.astype(np.float32)))
            return pj, ptj, z, jnp.zeros(K, jnp.int32), b

        out = {"status": "ok", "jax_version": jax.__version__, "devices": str(jax.devices())}
        # (1) vectors: the small golden case, first step
        p, pt, batches = G.fp_case_inputs("cnn_small")
        pj, ptj, z, c, b = run(p, pt, batches[0])
        _, _, _, _, l, g = jstep(pj, ptj, z, z, c, b, 6.25e-5, 1.5e-4, 0.99)
        out["cnn_small_losses"] = [float(x) for x in l]
        out["cnn_small_grad_probes"] = {n: [float(x) for x in np.asarray(gg).reshape(gg.shape[0], -1)[:, :8].reshape(-1)] for n, gg in g.items()}
        # (2) timing: the headline config
        cores = usable_cores()
        p = Q.init_params(0, "cnn", OBS, n_actions, FEATURES, K_HEADS)
        pt = Q.init_params(1, "cnn", OBS, n_actions, FEATURES, K_HEADS)
        s, a, r, s2, t = synthetic(0, n_actions)
        pj, ptj, m, c, b = run(p, pt, (s, a, r, s2, t))
        v = m
        pj, m, v, c, l, _ = jstep(pj, ptj, m, v, c, b, 6.25e-5, 1.5e-4, 0.99)
        jax.block_until_ready(l)
        n, t0 = 0, time.perf_counter()
        while True:
            pj, m, v, c, l, _ = jstep(pj, ptj, m, v, c, b, 6.25e-5, 1.5e-4, 0.99)
            jax.block_until_ready(l)
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or n >= 200:
                break
        out.update({"value": n / dt, "unit": "grad-steps/s", "cores": cores, "kind": "jax",
                    "sample": f"{n} jitted steps (vmap over K=5 heads, value_and_grad + Adam) in {dt:.1f} s on the host CPU"})
        return out
    except Exception as e:  # noqa: BLE001
        return {"status": f"harness error: {type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="profiling runs (rocprofv3 --pmc / --kernel-trace): nothing but the warm-up and the timed steps of THIS "
                         "configuration touches the GPU -- no per-launch event table, no sampling leg, no K-sweep")
    ap.add_argument("--heads-per-gpu", type=int, default=0,
                    help="head-parallel mode (BASELINE config 5, not the headline metric): K = this x gpus heads, every "
                         "rank steps its own window of heads on the same minibatch, no per-step collective; a T-step "
                         "and a D-step (neighbour exchange) are timed separately")
    ap.add_argument("--force-dp", action="store_true",
                    help="rehearse the data-parallel path (RCCL all-reduce + two-phase step) even with one rank")
    ap.add_argument("--actions", type=int, default=N_ACTIONS, help="action count A (SURVEY 8d: 6, repeat with 18)")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="ONE GPU: the compute side of rank 0 of an N-rank factored data-parallel step (the fused Dense_0 update "
                         "runs over N sample blocks: this rank's factors N times), no collective; a regression guard for the "
                         "part of weak scaling that does not depend on xGMI")
    ap.add_argument("--emulate-copies", choices=["serial", "side", "none"], default="serial",
                    help="--emulate-ranks: the N factor-block copies that stand in for the all-gather run on the compute stream "
                         "(serial: the tracked figure) or on a second stream under the conv backward, where the product path runs "
                         "its all-gather (side), or not at all (none: the gathered block is filled once before the timed region -- the "
                         "kernels' work does not depend on its contents)")
    ap.add_argument("--heads", default=str(K_HEADS),
                    help="heads K of the single-GPU step (default 5 = the headline; 64 = BASELINE config 5's single-device leg); a "
                         "comma list (1,2,3,5) prints ONLY the K-sweep fit: fixed_us + per_head_us * K")
    ap.add_argument("--batch", type=int, default=BATCH,
                    help="minibatch per GPU (default 32 = the headline; 256 = BASELINE config 4's per-step work on one device)")
    ap.add_argument("--dp-streams", choices=["side", "inline"], default=None,
                    help="data-parallel step (idqn_dp_step): collectives on the library's side stream (default) or on the compute stream")
    ap.add_argument("--learner", action="store_true",
                    help="the learner path the reference runs, end to end (idqn.py:65-72 = rb.sample() + learn_on_batch) on a 2^15-element "
                         "HBM frame ring: uniform sampler, prioritized sampler (reference protocol, 2^20-leaf tree) and the prioritized "
                         "learner with TD write-back (extension), B = 32 and 256, beside the bare step in the same run; its own JSON line")
    ap.add_argument("--algo", choices=["idqn", "iiqn"], default="idqn",
                    help="iiqn: BASELINE config 3 (i-IQN heads, 32 quantile fractions; a labelled extension -- the reference "
                         "has no quantile code), one GPU, its own JSON line")
    args = ap.parse_args()

    if (args.gpus > 1 or args.heads_per_gpu) and "WORLD_SIZE" not in os.environ:
        # No launcher around us: start the N ranks as fresh child processes (one per GPU, RCCL) and relay rank 0's
        # line.  Nothing in THIS process has touched torch, HIP or the extension yet, and it never will.
        import subprocess

        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
        lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if lines:
            print(lines[-1], flush=True)
        sys.exit(proc.returncode if proc.returncode or lines else 1)

    # Everything except the final JSON line goes to stderr -- including what native libraries print on fd 1
    # (RCCL writes a version banner to stdout at communicator creation).
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from collections import namedtuple

    from slimdqn import _hip

    # The extension normally travels prebuilt (graft build()); if this checkout has none, local rank 0 compiles it
    # (hipcc, ~1 min) before anything touches the GPU and the other ranks wait for the file.  No CPU path exists.
    if not os.path.exists(_hip.LIB_PATH):
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            import __graft_entry__

            __graft_entry__.build()
        else:
            for _ in range(600):
                if os.path.exists(_hip.LIB_PATH):
                    break
                time.sleep(1.0)
            time.sleep(2.0)  # let the linker finish writing
    from slimdqn.networks.idqn import iDQN
    from slimdqn.networks.parallel import data_parallel_step

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or args.force_dp or args.heads_per_gpu:
        os.environ.setdefault("MASTER_PORT", "29533")
        # the GPU boxes export NCCL_DEBUG=VERSION, so setdefault would never take: rank 0 logs RCCL's topology / algorithm
        # choice (stderr only; the JSON line goes to stdout), IDQN_NCCL_DEBUG overrides
        os.environ["NCCL_DEBUG"] = os.environ.get("IDQN_NCCL_DEBUG", "INFO" if rank == 0 else "WARN")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
        if rank == 0:  # rank 0's RCCL log also goes to a file that the JSON line quotes from (rccl.debug_excerpt)
            os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/idqn_rccl_rank0_{os.getpid()}.log")
        assert world == args.gpus, f"--gpus {args.gpus} needs torchrun with {args.gpus} ranks (WORLD_SIZE={world})"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    Batch = namedtuple("Batch", "state action reward next_state is_terminal")
    if "," in args.heads:  # the K-sweep alone
        fit = heads_fit(args.actions, tuple(int(x) for x in args.heads.split(",")), steps=min(args.steps, 300))
        os.write(json_fd, (json.dumps({"metric": "i-DQN step time by heads K (B = 32), line fit", "unit": "us", "n_gpus": 1, **fit}) + "\n").encode())
        return
    K, B = int(args.heads), int(args.batch)
    headline = K == K_HEADS and B == BATCH
    if args.heads_per_gpu:
        return head_parallel_bench(args, rank, world, json_fd, Batch)
    if args.algo == "iiqn":
        return iiqn_bench(args, json_fd, Batch)
    if args.emulate_ranks:
        return emulate_ranks_bench(args, json_fd, Batch)
    if args.learner:
        return learner_bench(args, json_fd, Batch)
    A = args.actions
    dp = world > 1 or args.force_dp
    # The CPU leg FIRST (the reference path is timed before the GPU runs), rank 0 of a single-GPU run only.
    cpu = None
    if rank == 0 and not dp and not args.no_cpu_baseline:
        cpu = cpu_baseline(n_actions=A, heads=K, batch=B)
        cpu["jax"] = jax_cpu_probe(n_actions=A) if headline else {"status": "probed on the headline configuration only"}
    agent = iDQN(0, OBS, A, K, FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
    # 8 distinct synthetic minibatches per rank, resident in HBM before the timed region, used round-robin
    batches = [Batch(*(torch.from_numpy(x).cuda() for x in synthetic(1000 + 64 * rank + i, A, B))) for i in range(8)]
    it = [0]
    global_batch = B * world
    dp_mode = (os.environ.get("IDQN_DP_MODE", "native") if dp else "single")
    if dp and rank == 0:  # the first multi-GPU run should show which algorithm / protocol RCCL picks over xGMI
        print(f"[bench] NCCL_DEBUG={os.environ.get('NCCL_DEBUG')} IDQN_DP_MODE={dp_mode} streams={args.dp_streams or os.environ.get('IDQN_DP_STREAMS', 'side')}",
              file=sys.stderr, flush=True)

    def step(flags=0):
        batch = batches[it[0] % len(batches)]
        it[0] += 1
        if not dp:
            agent._learn(batch, flags=flags)
        else:
            data_parallel_step(agent, batch, global_batch, extra_flags=flags, mode=dp_mode, streams=args.dp_streams)

    def barrier():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    # data-parallel runs: the SAME rank's plain single-GPU step (no collective, no factor exchange) timed first, so that the line
    # can say what N ranks cost relative to one -- t(1) / t(N) at fixed per-rank work -- instead of leaving it to a second run
    single_ms = None
    if dp:
        for _ in range(20):
            agent._learn(batches[0])
        t1s = []
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            for i in range(100):
                agent._learn(batches[i % len(batches)])
            torch.cuda.synchronize()
            t1s.append((time.perf_counter() - t0) / 100)
        t1 = torch.tensor([float(np.median(t1s))], dtype=torch.float64, device="cuda")
        dist.all_reduce(t1, op=dist.ReduceOp.MAX)
        single_ms = float(t1.item()) * 1e3

    for _ in range(args.warmup):
        step()

    # `repeats` timed regions of EXACTLY `steps` steps each, barrier + synchronize on both sides, max over ranks;
    # the reported region is the median one.  Short regions (the driver's --steps 20 is 5.6 ms of GPU time) are repeated until
    # ~1500 steps have been timed: the first regions after the CPU leg still see the clocks ramp (the round-4 driver line read
    # 0.2946, 0.2854, 0.2816, 0.2789, 0.2763 ms over its five) and a single host hiccup is a whole region; the median over more
    # regions is the steady-state figure SURVEY 8d defines the metric on.  Every region is listed under timing.ms_per_step_all.
    import gc

    gc.collect()
    n_regions = max(1, args.repeats, min(75, -(-1500 // max(1, args.steps)))) | 1  # odd: the median is a region that was run
    regions = []
    for _ in range(n_regions):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            # the dominant kernel is bracketed with hipEvents on every PROFILE_EVERY-th step of the region only: a bracketed
            # launch cannot overlap its neighbours' launch / tail phases, which costs the step 4.5 - 5 us (measured: the fifth
            # 500-step region of the earlier all-steps scheme ran past the 2048-event buffer and was that much faster)
            step(_hip.F_PROFILE if i % PROFILE_EVERY == 0 else 0)
        barrier()
        elapsed = time.perf_counter() - t0
        if dp:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        regions.append(elapsed)
    elapsed = float(np.median(regions))

    import ctypes as C

    mean_ms, n_l, name = C.c_double(), C.c_int32(), C.create_string_buffer(64)
    _hip.check(_hip.lib().idqn_profile_read(agent._handle, C.byref(mean_ms), C.byref(n_l), name), "idqn_profile_read")
    # per-launch table: a short extra run (outside the timed regions) with one hipEvent behind every launch
    kernels = []
    if not dp and not args.no_side_legs:
        for _ in range(60):
            step(_hip.F_PROFILE_ALL)
        torch.cuda.synchronize()
        buf = C.create_string_buffer(8192)
        _hip.check(_hip.lib().idqn_profile_table(agent._handle, buf, 8192), "idqn_profile_table")
        for ln in buf.value.decode().splitlines():
            nm, us, cnt = ln.split("\t")
            kernels.append({"launch": nm, "us": float(us), "n": int(cnt)})
    losses = agent._losses.cpu().numpy()
    assert np.isfinite(losses).all(), losses
    if dp_mode == "native" and getattr(agent, "_native_dp_failed", False):
        dp_mode = "factored"  # (the library's communicator did not come up on every rank: parallel.py ran its Python schedule)
    if rank == 0:
        P_w0 = 7744 * 512
        fused = dp_mode != "allreduce"
        # algorithmic HBM bytes of one launch of the dominant kernel:
        # fused: theta, m, v of Dense_0/kernel read + written; unfused: gradient written; + a3 and dh read once per sample block
        # (data-parallel: the factors of all `world` ranks)
        nb = -(-B // 32)
        n_blocks = nb * (world if fused else 1)
        per_head = (6 if fused else 1) * P_w0 * 4 + n_blocks * (7744 * 32 * 4 + 512 * 32 * 4)
        if not dp and nb < 3:
            per_head += nb * 7744 * 32 * 4  # single-device path up to two sample blocks: the kernel also emits dL/da3 (counted once)
        alg_bytes = K * per_head
        achieved = alg_bytes / (mean_ms.value * 1e-3) / 1e9 if mean_ms.value > 0 else 0.0
        # HBM bytes from the committed PMC passes (tools/gpu_pmc.sh -> tools/pmc_summarise.py): per launch of the dominant
        # kernel, and summed over the launches of one step (headline configuration only: that is what the passes ran)
        traffic, traffic_source, step_traffic = None, None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
            if headline and not dp:
                # the passes are a separate run (counters cannot ride on the timed run): their figures stand only for the kernel
                # sources they were collected on -- any other csrc/ digest and the line says so instead of quoting them
                if pmc.get("csrc_sha256") == csrc_digest():
                    traffic = pmc["hbm_bytes_per_launch"]
                    step_traffic = pmc.get("step_hbm_bytes")
                    traffic_source = (f"profiles/pmc_traffic_latest.json (separate rocprofv3 --pmc passes on these kernel sources, csrc digest "
                                      f"{pmc['csrc_sha256'][:12]}, library {pmc.get('git', '?')}; not this run)")
                else:
                    traffic_source = (f"null: profiles/pmc_traffic_latest.json was collected on other kernel sources (csrc digest "
                                      f"{str(pmc.get('csrc_sha256'))[:12]} vs {csrc_digest()[:12]} here): re-run tools/gpu_pmc.sh + tools/pmc_summarise.py")
        except Exception as e:  # noqa: BLE001
            traffic_source = f"null: {type(e).__name__}: {e}"
        step_flops, step_bytes = step_work(K, B, A)
        ms_step = elapsed / args.steps * 1e3
        out = {
            "metric": f"i-DQN grad-steps/sec, Nature-CNN K={K} batch={B}",
            "value": args.steps * world / elapsed,
            "unit": "grad-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Atari synthetic 84x84x4 uint8, i-DQN K={K} Nature-CNN [32,64,64,512] A={A}, "
                                   f"batch {B} per GPU (global {global_batch}), "
                                   + {"single": "fused wgrad+Adam",
                                      "native": "ONE C call per step (idqn_dp_step): Dense_0 factors all-gathered (RCCL), fused "
                                                "wgrad+Adam over the global batch, small leaves all-reduced",
                                      "factored": "the same schedule issued from Python over torch.distributed",
                                      "allreduce": "grad all-reduce (RCCL) then Adam"}[dp_mode],
                       "heads": K, "batch_per_gpu": B, "global_batch": global_batch, "actions": A,
                       "parallelism": f"dp{world}" if dp else "single",
                       "conv_arithmetic": os.environ.get("IDQN_CONV", "bf16x3") + " (f32-accurate products)"},
            "timing": {"regions": n_regions, "steps_per_region": args.steps, "reported": "median region",
                       "ms_per_step_all": [r / args.steps * 1e3 for r in regions]},
            "roofline": {"bound": "hbm", "kernel": name.value.decode() + ("<fused Adam>" if fused else "<grad only>"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "launch_ms": mean_ms.value, "launches_timed": n_l.value, "timed_every": PROFILE_EVERY, "algorithmic_bytes": alg_bytes,
                         # context, not the contract's `frac`: what a pure non-temporal copy reaches on this chip
                         # (profiles/r4_ldsdma_copy_probe.txt: 6.1-6.25 TB/s read + write; MI355X_MICROARCH.md: 6.29)
                         "copy_ceiling": 6200.0, "frac_of_copy_ceiling": achieved / 6200.0},
            # the whole step against its two floors (per GPU: every rank does one 32-sample step per global step)
            "step_roofline": {"flops": step_flops, "bytes": step_bytes,
                              "floor_us_mfma_f32": step_flops / MFMA_F32_PEAK * 1e6, "floor_us_hbm": step_bytes / (HBM_PEAK_GBS * 1e9) * 1e6,
                              "frac_mfma": step_flops / MFMA_F32_PEAK / (ms_step * 1e-3),
                              "frac_hbm": step_bytes / (HBM_PEAK_GBS * 1e9) / (ms_step * 1e-3),
                              # HBM bytes of ALL launches of one step from the PMC passes against the algorithmic bytes
                              "traffic": step_traffic, "traffic_over_algorithmic": (step_traffic / step_bytes) if step_traffic else None,
                              "peaks": "157.3 TFLOP/s f32-rate MFMA, 8 TB/s HBM3E (MI355X_MICROARCH.md)"},
            "kernels": kernels,
            "final_losses": [float(x) for x in losses],
        }
        if dp:
            out["rccl"] = rccl_report(dp_mode, world, agent, args.dp_streams or os.environ.get("IDQN_DP_STREAMS", "side"))
            # `value` counts every rank's 32-sample step (weak scaling: N ranks = N x the samples per global step), so it grows
            # with N by definition.  What N ranks buy is in these four, none of which does:
            out["data_parallel"] = {
                "per_rank_ms_per_step": ms_step,                       # one global step = one step on every rank, max over ranks
                "global_steps_per_s": 1e3 / ms_step,                   # optimizer updates per second (global batch B x N each)
                "samples_per_s": global_batch * 1e3 / ms_step,
                "rank_steps_per_s": args.steps * world / elapsed,      # = value
                "single_gpu_ms_per_step": single_ms,                   # the plain step on the same ranks, same run (max over ranks)
                "scaling_efficiency": (single_ms / ms_step) if single_ms else None,  # t(1) / t(N) at fixed per-rank work
                "what": "weak scaling: per-rank batch fixed at %d, global batch %d; efficiency 1.0 = the N-rank step costs what the "
                        "single-GPU step costs" % (B, global_batch)}
        if headline and not dp and not args.no_side_legs:
            try:
                out["sampling"] = sampling_leg()
            except Exception as e:  # noqa: BLE001 -- the headline line must not depend on this leg
                out["sampling"] = {"error": f"{type(e).__name__}: {e}"}
            try:  # the K-independent part of the step, tracked
                out["heads_fit"] = heads_fit(A)
            except Exception as e:  # noqa: BLE001
                out["heads_fit"] = {"error": f"{type(e).__name__}: {e}"}
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = out["value"] / cpu["value"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dp:
        agent._destroy_handle()  # (the library-side communicator goes before the process group)
        dist.destroy_process_group()


def learner_bench(args, json_fd, Batch):
    """`update_online_params` in a loop (idqn.py:65-72: `replay_buffer.sample()` + `learn_on_batch`, update_to_data = 1) against
    the bare step on pre-resident batches, same process, same box, regions interleaved.  Buffers hold 2^15 elements of
    Atari-shaped frames; the prioritized samplers' trees have 2^20 leaves (BASELINE config 4).  One JSON line."""
    import torch

    from slimdqn.networks.idqn import iDQN
    from slimdqn.sample_collection.per import PrioritizedLearner, SlotPrioritizedSampler
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
    from slimdqn.sample_collection.samplers import PrioritizedSamplingDistribution, UniformSamplingDistribution

    A, K = args.actions, int(args.heads)
    n_el, leaves = 1 << 15, 1 << 20
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 256, (512, 84, 84), dtype=np.uint8)  # (cycled: the loop's cost does not depend on the pixels)
    acts, rews = rng.integers(0, A, n_el + 8), rng.integers(-1, 2, n_el + 8)

    def fill(rb, **kw):
        for i in range(n_el + 4):
            rb.add(TransitionElement(frames[i % 512], int(acts[i]), float(rews[i]), bool(i % 1000 == 999), False), **kw)
        rb.reuse_sample_buffers = True  # as the trainer sets it (experiments/base/launch.py): a batch is consumed before the next is drawn

    out = {"metric": f"i-DQN learner loop: replay_buffer.sample() + learn_on_batch, Nature-CNN K={K}", "unit": "grad-steps/s", "n_gpus": 1,
           "data": "synthetic", "dtype": "f32", "higher_is_better": True,
           "config": {"workload": f"update_online_params (idqn.py:65-72) on a 2^15-element HBM frame ring of 84x84 uint8 frames, stack 4, "
                                  f"K={K} A={A}; prioritized trees 2^20 leaves", "replay_elements": n_el, "tree_leaves": leaves},
           "legs": {"bare": "agent._learn on 8 pre-resident batches (what bench.py's headline times)",
                    "uniform": "UniformSamplingDistribution: host PCG64 draw + index map, then ONE C call (idqn_learn_on_replay: slots as "
                               "kernel arguments, the stacked gather inside the step's staging launch)",
                    "prioritized": "PrioritizedSamplingDistribution (samplers.py:52-116), no write-back (the reference's ReplayBuffer.sample drops "
                                   "the keys): query on the tree's own stream + polled mailbox, keys to the host, then idqn_learn_on_replay",
                    "uniform_two_calls / prioritized_two_calls": "the same with sample() and learn_on_batch as two calls: slots uploaded, "
                                                                 "replay_gather_stacked launch, idqn_learn_on_batch on its outputs",
                    "prioritized_learner": "PrioritizedLearner (extension): device-side stratified sample, importance weights, gather, "
                                           "step, |TD| -> priorities -> sumtree_set, no host read"}}
    steps, reps = max(50, min(args.steps, 300)), 3
    for B in (32, 256):
        agent = iDQN(0, OBS, A, K, FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
        batches = [Batch(*(torch.from_numpy(x).cuda() for x in synthetic(2000 + i, A, B))) for i in range(8)]
        rb_u = ReplayBuffer(UniformSamplingDistribution(1), batch_size=B, max_capacity=n_el, stack_size=4, update_horizon=1, gamma=0.99)
        fill(rb_u)
        rb_p = ReplayBuffer(PrioritizedSamplingDistribution(2, leaves), batch_size=B, max_capacity=n_el, stack_size=4, update_horizon=1, gamma=0.99)
        pri = rng.random(n_el + 8) + 0.01
        for i in range(n_el + 4):
            rb_p.add(TransitionElement(frames[i % 512], int(acts[i]), float(rews[i]), bool(i % 1000 == 999), False), priority=float(pri[i]))
        rb_p.reuse_sample_buffers = True
        rb_l = ReplayBuffer(SlotPrioritizedSampler(3, leaves, priority_exponent=0.6), batch_size=B, max_capacity=leaves, stack_size=4,
                            update_horizon=1, gamma=0.99)
        fill(rb_l)
        learner = PrioritizedLearner(agent, rb_l, beta=0.4, eps=1e-3, reduce="mean")
        it = [0]

        def bare():
            agent._learn(batches[it[0] % 8])
            it[0] += 1

        def two_calls(rb):  # sample() then learn_on_batch, as two calls (what the fused call replaces)
            def fn():
                agent.fuse_replay_sampling = False
                agent.update_online_params(0, rb)
                agent.fuse_replay_sampling = True
            return fn

        legs = {"bare": bare, "uniform": lambda: agent.update_online_params(0, rb_u),
                "prioritized": lambda: agent.update_online_params(0, rb_p),
                "uniform_two_calls": two_calls(rb_u), "prioritized_two_calls": two_calls(rb_p), "prioritized_learner": learner.step}
        times = {n: [] for n in legs}
        for fn in legs.values():
            for _ in range(20):
                fn()
        torch.cuda.synchronize()
        for _ in range(reps):  # interleaved regions: every leg sees the same clocks
            for name, fn in legs.items():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    fn()
                torch.cuda.synchronize()
                times[name].append((time.perf_counter() - t0) / steps)
        # host-side cost of one sample() alone (no step behind it): what has to hide under the step in flight
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            rb_p.sample()
        host_p = (time.perf_counter() - t0) / 100
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            rb_u.sample()
        host_u = (time.perf_counter() - t0) / 100
        torch.cuda.synchronize()
        res = {}
        bare_ms = float(np.median(times["bare"])) * 1e3
        for name in legs:
            ms = float(np.median(times[name])) * 1e3
            res[name] = {"ms_per_step": ms, "steps_per_s": 1e3 / ms, "vs_bare": bare_ms / ms, "regions_ms": [x * 1e3 for x in times[name]]}
        res["sample_call_host_us"] = {"uniform": host_u * 1e6, "prioritized": host_p * 1e6,
                                      "what": "wall clock of rb.sample() alone, issued back to back with nothing else queued"}
        out[f"B{B}"] = res
        assert np.isfinite(agent._losses.cpu().numpy()).all()
        del agent, rb_u, rb_p, rb_l, learner
        torch.cuda.empty_cache()
    out["value"] = out["B32"]["uniform"]["steps_per_s"]
    out["ms_per_step"] = out["B32"]["uniform"]["ms_per_step"]
    out["steps"] = steps
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def iiqn_cpu_baseline(n_actions, n_quantiles, budget_s=12.0):
    """The oracle's torch-CPU fp32 restatement of the i-IQN step (K heads one after the other: loss, autograd, Adam)."""
    import torch

    from oracle import iqn_ref as I
    from oracle import qnet_ref as Q
    from oracle import torch_ref as T

    cores = usable_cores()
    torch.set_num_threads(cores)
    p = I.init_params(0, OBS, n_actions, FEATURES, K_HEADS)
    pt = I.init_params(1, OBS, n_actions, FEATURES, K_HEADS)
    mu = {n: np.zeros_like(a) for n, a in p.items()}
    nu = {n: np.zeros_like(a) for n, a in p.items()}
    s, a, r, s2, t = synthetic(0, n_actions)
    batch = (s, a, r, s2, t.astype(bool))
    taus = I.synthetic_taus(1, K_HEADS, n_quantiles, BATCH)

    def step(count):
        for k in range(K_HEADS):
            _, g = T.iqn_loss_and_grads(Q.head(p, k), Q.head(pt, k), batch, tuple(taus[k]), 0.99, dtype=torch.float32)
            for n in p:
                p[n][k], mu[n][k], nu[n][k] = Q.adam_update(p[n][k], g[n], mu[n][k], nu[n][k], count, 6.25e-5, 1.5e-4, np.float32)

    step(0)
    n, t0 = 0, time.perf_counter()
    while True:
        step(n + 1)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 50:
            break
    return {"value": n / dt, "unit": "grad-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of the same K=5 B=32 N={n_quantiles} i-IQN step in {dt:.1f} s (oracle/torch_ref.iqn_loss_and_grads, "
                      f"torch-CPU fp32 autograd + numpy Adam, {cores} threads; no reference implementation exists)"}


def iiqn_bench(args, json_fd, Batch):
    """BASELINE config 3: K=5 i-IQN heads, 32 quantile fractions (online / action selection / target), batch 32, one GPU."""
    import ctypes as C

    import torch

    from slimdqn import _hip
    from slimdqn.networks.iiqn import iIQN

    A, N = args.actions, 32
    cpu = None if args.no_cpu_baseline else iiqn_cpu_baseline(A, N)  # the CPU leg first
    agent = iIQN(0, OBS, A, K_HEADS, FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4, n_quantiles=N)
    batches = [Batch(*(torch.from_numpy(x).cuda() for x in synthetic(1000 + i, A))) for i in range(8)]
    it = [0]

    def step(flags=0):
        agent._learn(batches[it[0] % 8], flags=flags)  # draws the step's fractions on the host and uploads them (60 KB)
        it[0] += 1

    steps = min(args.steps, 100)
    for _ in range(min(args.warmup, 10)):
        step()
    regions = []
    for _ in range(max(1, args.repeats)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        regions.append(time.perf_counter() - t0)
    elapsed = float(np.median(regions))
    for _ in range(10):
        step(_hip.F_PROFILE_ALL)
    torch.cuda.synchronize()
    buf = C.create_string_buffer(8192)
    _hip.check(_hip.lib().idqn_profile_table(agent._handle, buf, 8192), "idqn_profile_table")
    kernels = []
    for ln in buf.value.decode().splitlines():
        nm, us, cnt = ln.split("\t")
        kernels.append({"launch": nm, "us": float(us), "n": int(cnt)})
    losses = agent._losses.cpu().numpy()
    assert np.isfinite(losses).all(), losses
    F, J = 7744, 512
    rows = N * BATCH  # (sample, fraction) rows per virtual net
    # FLOPs of the contraction-bound launches (i-IQN section).  The Dense_0 GEMMs (csrc/iqn_gemm.h) issue six bf16
    # products per f32 product (the exact three-plane split): priced on the bf16 MFMA peak with 6x the f32-equivalent count;
    # likewise the embedding forward and backward (k_iqn_embed3, k_iqn_embed_bwd3).
    gemm = 2.0 * K_HEADS * rows * F * J
    work = {
        "iqn dense0 fwd": (3 * gemm, 6, MFMA_BF16_PEAK),
        "iqn dense0 dgrad": (gemm, 6, MFMA_BF16_PEAK),
        "iqn dense0 wgrad": (gemm, 6, MFMA_BF16_PEAK),
        "iqn dense0 dgrad + wgrad": (2 * gemm, 6, MFMA_BF16_PEAK),
        "iqn embedding x features": (2.0 * 3 * K_HEADS * rows * 64 * F, 6, MFMA_BF16_PEAK),
        "iqn embedding backward": (2.0 * 2 * K_HEADS * rows * 64 * F, 6, MFMA_BF16_PEAK),
    }
    for kr in kernels:  # per-launch MFMA rate beside the time
        if kr["launch"] in work:
            f32eq, mult, peak = work[kr["launch"]]
            kr["mfma_tflops"] = f32eq * mult / (kr["us"] * 1e-6) / 1e12  # ISSUED bf16 products: what the matrix pipe does
            kr["mfma_frac"] = kr["mfma_tflops"] / (peak / 1e12)
            kr["f32eq_tflops"] = f32eq / (kr["us"] * 1e-6) / 1e12       # USEFUL f32-equivalent work (issued / mult)
    dom = max(kernels, key=lambda k: k["us"]) if kernels else None
    roof = None
    if dom and dom["launch"] in work:
        f32eq, mult, peak = work[dom["launch"]]
        ach = f32eq * mult / (dom["us"] * 1e-6) / 1e12
        roof = {"bound": "mfma", "kernel": dom["launch"], "achieved": ach, "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": ach / (peak / 1e12), "traffic": None, "launch_ms": dom["us"] * 1e-3,
                # `achieved` / `frac` price the ISSUED bf16 products (pipe utilisation); the useful f32-equivalent work is
                # 1 / mult of that -- both are stated so that nobody reads `frac` as useful-FLOP efficiency
                "frac_issued": ach / (peak / 1e12), "achieved_algorithmic": f32eq / (dom["us"] * 1e-6) / 1e12,
                "frac_algorithmic_vs_f32_mfma_peak": f32eq / (dom["us"] * 1e-6) / MFMA_F32_PEAK,
                "algorithmic_flops": f32eq, "issued_flops": f32eq * mult,
                "note": ("f32-accurate contraction as six bf16 products per f32 product: issued bf16 FLOPs (6 x algorithmic) against "
                         "the dense bf16 MFMA peak at 2.4 GHz; measured in-kernel clock of this launch 2.06 GHz (tools/probes/iqn_clock.py)" if mult == 6 else
                         "f32 MFMA FLOPs against the f32 MFMA peak")}
    out = {"metric": "i-IQN grad-steps/sec, Nature-CNN K=5 batch=32, 32 quantile samples", "value": steps / elapsed,
           "unit": "grad-steps/s", "n_gpus": 1, "steps": steps, "warmup": min(args.warmup, 10), "ms_per_step": elapsed / steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"Atari synthetic 84x84x4 uint8, i-IQN K=5 Nature-CNN [32,64,64,512] A={A}, batch 32, "
                                  f"{N} online / {N} action-selection / {N} target fractions per sample (extension: the reference "
                                  "has no quantile code; parity pinned to oracle/iqn_ref.py only)",
                      "heads": K_HEADS, "batch_per_gpu": BATCH, "quantiles": N, "actions": A},
           "timing": {"regions": args.repeats, "steps_per_region": steps, "reported": "median region",
                      "ms_per_step_all": [r / steps * 1e3 for r in regions]},
           "roofline": roof, "kernels": kernels, "final_losses": [float(x) for x in losses]}
    if cpu is not None:
        out["cpu_baseline"] = cpu
        out["gpu_over_cpu"] = out["value"] / cpu["value"]
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def emulate_ranks_bench(args, json_fd, Batch):
    """Rank 0's compute of an N-rank factored step on one GPU (slimdqn/networks/parallel.py issues exactly these calls; the
    all-gather is replaced by N copies of this rank's own factors)."""
    import ctypes as C  # noqa: F401

    import torch

    from slimdqn import _hip
    from slimdqn.networks.idqn import iDQN

    N, A = args.emulate_ranks, args.actions
    agent = iDQN(0, OBS, A, K_HEADS, FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
    batches = [Batch(*(torch.from_numpy(x).cuda() for x in synthetic(1000 + i, A))) for i in range(8)]
    lib, q = _hip.lib(), _hip.current_stream
    K = agent._K
    F, J = next(shape for name, _, shape in agent._leaves if name == "Dense_0/kernel")
    X, Y = F * 32, J * 32
    n_a3, n_dh = K * X, K * Y
    send = torch.zeros(n_a3 + n_dh, dtype=torch.float32, device="cuda")
    gathered = torch.zeros(N * (n_a3 + n_dh), dtype=torch.float32, device="cuda")
    it = [0]
    side = torch.cuda.Stream() if args.emulate_copies == "side" else None

    def step():
        agent._learn(batches[it[0] % 8], flags=_hip.F_STOP_BEFORE_DENSE0_WGRAD, mean_divisor=32 * N)
        it[0] += 1
        _hip.check(lib.idqn_export_dense0_factors(agent._handle, _hip.ptr(send), _hip.ptr(send[n_a3:]), q()), "export")
        if args.emulate_copies == "none" and it[0] > 1:
            pass
        elif side is None:
            for r in range(N):  # (stands for the all-gather: every slot holds this rank's factors)
                gathered[r * (n_a3 + n_dh) : (r + 1) * (n_a3 + n_dh)].copy_(send)
        else:  # as the product path runs its all-gather: on a second stream, under the conv backward
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for r in range(N):
                    gathered[r * (n_a3 + n_dh) : (r + 1) * (n_a3 + n_dh)].copy_(send)
        _hip.check(lib.idqn_backward_rest(agent._handle, q()), "rest")
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        fa = (agent._handle, _hip.ptr(gathered), _hip.ptr(gathered[n_a3:]), N, 1, n_a3 + n_dh, X, X, n_a3 + n_dh, Y, Y)
        _hip.check(lib.idqn_finish_step_factored(*fa, _hip.FACTORED_DENSE0, q()), "finish")
        _hip.check(lib.idqn_finish_step_factored(*fa, _hip.FACTORED_REST, q()), "finish")

    for _ in range(args.warmup):
        step()
    regions = []
    for _ in range(max(1, args.repeats)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        regions.append(time.perf_counter() - t0)
    elapsed = float(np.median(regions))
    out = {"metric": "compute side of one rank of an N-rank factored i-DQN step (no collective), steps/s", "emulated_ranks": N,
           "value": args.steps / elapsed, "unit": "rank-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "copies": args.emulate_copies,
           "note": f"includes {N} device copies of the 5.3 MB factor block standing in for the all-gather "
                   f"({'none inside the timed region' if args.emulate_copies == 'none' else 'on the compute stream' if side is None else 'on a second stream under the conv backward, as the all-gather runs'}); "
                   f"the Dense_0 update contracts {N} sample blocks per head",
           "final_losses": [float(x) for x in agent._losses.cpu().numpy()]}
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def rccl_report(dp_mode, world, agent, streams="side"):
    """What the first multi-GPU run should explain by itself: bytes handed to each collective per step, and RCCL's own
    lines about topology / algorithm / channels (NCCL_DEBUG_FILE of rank 0, set before the communicator is created)."""
    K = agent._K
    rep = {"mode": dp_mode, "world": world, "collectives_per_step": [],
           "issued_by": ("idqn_dp_step (csrc/dp.hip): ncclAllGather / ncclAllReduce on the library's own communicator, "
                         + ("a side stream ordered by hipEvents" if streams == "side" else "the compute stream")) if dp_mode == "native"
                        else "slimdqn/networks/parallel.py over torch.distributed (backend nccl = RCCL)"}
    try:
        F, J = next(shape for name, _, shape in agent._leaves if name == "Dense_0/kernel")
        small = int(agent._grad_small.numel()) * 4
        if dp_mode in ("factored", "native"):
            per_rank = K * (F + J) * 32 * 4
            rep["collectives_per_step"] = [
                {"op": "all_gather", "what": "Dense_0 gradient factors a3 | dh", "bytes_sent_per_rank": per_rank,
                 "bytes_received_per_rank": per_rank * (world - 1)},
                {"op": "all_reduce", "what": "small-leaf gradients + K losses", "bytes": small}]
        else:
            rep["collectives_per_step"] = [
                {"op": "all_reduce", "what": "Dense_0/kernel gradients", "bytes": int(agent._grad_w0.numel()) * 4},
                {"op": "all_reduce", "what": "small-leaf gradients + K losses", "bytes": small}]
    except Exception as e:  # noqa: BLE001
        rep["error"] = f"{type(e).__name__}: {e}"
    try:
        path = os.environ.get("NCCL_DEBUG_FILE", "")
        if path and os.path.exists(path):
            keep = ("Channel", "Ring", "Tree", "Using", "algo", "proto", "nChannels", "XGMI", "P2P", "NET/", "comm 0x")
            lines = [ln.strip() for ln in open(path, errors="replace") if any(k in ln for k in keep)]
            rep["debug_excerpt"] = lines[:40]
            rep["debug_lines"] = len(lines)
    except Exception as e:  # noqa: BLE001
        rep["debug_error"] = f"{type(e).__name__}: {e}"
    return rep


def head_parallel_bench(args, rank, world, json_fd, Batch):
    """BASELINE config 5: K = heads_per_gpu x world heads, the same minibatch on every rank, no per-step collective."""
    import torch
    import torch.distributed as dist

    from slimdqn.networks.head_parallel import HeadShardedIDQN

    K = args.heads_per_gpu * world
    agent = HeadShardedIDQN(0, OBS, args.actions, K, FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
    batches = [Batch(*(torch.from_numpy(x).cuda() for x in synthetic(1000 + i, args.actions))) for i in range(8)]  # same on all ranks

    def timed(fn, n):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / n

    for i in range(args.warmup):
        agent._learn(batches[i % 8])
    step_s = timed(lambda i: agent._learn(batches[i % 8]), args.steps)
    agent._target_update()
    agent._target_sync()
    t_s = timed(lambda i: agent._target_update(), 20)
    d_s = timed(lambda i: agent._target_sync(), 20)
    assert np.isfinite(agent._losses.cpu().numpy()).all()
    if rank == 0:
        out = {"metric": "i-DQN head-gradient-steps/sec, Nature-CNN batch=32, head-parallel (BASELINE config 5)",
               "value": K / step_s, "unit": "head-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"i-DQN K={K} ({args.heads_per_gpu} heads per GPU), batch 32 replicated, "
                                      "no per-step collective", "heads": K, "parallelism": f"hp{world}"},
               "target_update_ms": t_s * 1e3, "target_sync_ms": d_s * 1e3}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
