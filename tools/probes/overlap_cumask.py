"""Does the HBM-bound Dense_0 update overlap with the MFMA/latency-bound conv launches when each gets its own part of the chip?

No kernel changes: the step is issued in the pieces the factored data-parallel path already has
(IDQN_F_STOP_BEFORE_DENSE0_WGRAD -> idqn_backward_rest / idqn_finish_step_factored), the fused Dense_0 update goes to a
second stream created with hipExtStreamCreateWithCUMask (n_b CUs), everything else stays on the caller's stream with its
conv launches planned for IDQN_CUS = 256 - n_b workgroups.  Modes:
  fused     the shipped single-stream step (one C call)
  serial    the split step, all on one stream (what the split itself costs: separate data gradient, slab reduce)
  within    fork after the data gradient, join before the small-leaf Adam (same step)
  (round 3 also had `cross`: join in front of the NEXT step's Dense_0 forward through an experiment hook in the library;
   its figures are in profiles/r3_overlap_cumask_streams.txt, the hook was removed in round 4: the .so exports only what
   include/idqn_hip.h declares)
usage: python tools/probes/overlap_cumask.py <n_b> <pattern: low|stride|none> [steps]
"""
import os
os.environ.setdefault("IDQN_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "i-dqn_amd", "libidqn_hip_variants.so"))  # variants build: the stamps / switches used here
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "i-dqn_amd")):
    sys.path.insert(0, p)

n_b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pattern = sys.argv[2] if len(sys.argv) > 2 else "low"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from slimdqn import _hip  # noqa: E402
from slimdqn.networks.idqn import iDQN  # noqa: E402
from collections import namedtuple  # noqa: E402

hip = C.CDLL("libamdhip64.so")
lib = _hip.lib()
for _f, _a in (("hipEventRecord", [C.c_void_p, C.c_void_p]), ("hipStreamWaitEvent", [C.c_void_p, C.c_void_p, C.c_uint]),
               ("hipStreamSynchronize", [C.c_void_p]), ("hipEventSynchronize", [C.c_void_p]),
               ("hipEventElapsedTime", [C.POINTER(C.c_float), C.c_void_p, C.c_void_p])):
    getattr(hip, _f).argtypes = _a
torch.cuda.set_device(0)


def chk(e, what):
    assert e == 0, f"{what} failed: {e}"


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    chk(hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words), "hipExtStreamCreateWithCUMask")
    return s


def event(timing=False):
    e = C.c_void_p()
    chk(hip.hipEventCreateWithFlags(C.byref(e), 0 if timing else 2), "hipEventCreateWithFlags")
    return e


if pattern == "none":
    sB = C.c_void_p()
    chk(hip.hipStreamCreateWithFlags(C.byref(sB), 1), "hipStreamCreateWithFlags")
else:
    if pattern == "low":
        bits = list(range(n_b))
    elif pattern == "high":
        bits = list(range(256 - n_b, 256))
    else:
        stride = 256 // n_b
        bits = [i * stride for i in range(n_b)]
    sB = masked_stream(bits)

Batch = namedtuple("Batch", "state action reward next_state is_terminal")
agent = iDQN(0, bench.OBS, 6, bench.K_HEADS, bench.FEATURES, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
batches = [Batch(*(torch.from_numpy(x).cuda() for x in bench.synthetic(1000 + i))) for i in range(8)]
q = _hip.current_stream
K = agent._K
F, J = next(shape for name, _, shape in agent._leaves if name == "Dense_0/kernel")
X, Y = F * 32, J * 32
e_fork, e_join = event(), event()
t0e, t1e = event(True), event(True)


def factors():
    p, c_dh, c_a3 = C.c_void_p(), C.c_int64(), C.c_int64()
    _hip.check(lib.idqn_dense0_factors(agent._handle, C.byref(p), C.byref(c_dh), C.byref(c_a3)), "idqn_dense0_factors")
    dh = p.value
    a3 = dh + 4 * c_dh.value
    return (agent._handle, C.c_void_p(a3), C.c_void_p(dh), 1, 1, 0, X, X, 0, Y, Y)


def step_fused(i):
    agent._learn(batches[i % 8])


def step_split(i, mode):
    agent._learn(batches[i % 8], flags=_hip.F_STOP_BEFORE_DENSE0_WGRAD)
    args = factors()
    if mode == "serial":
        _hip.check(lib.idqn_backward_rest(agent._handle, q()), "rest")
        _hip.check(lib.idqn_finish_step_factored(*args, _hip.FACTORED_DENSE0, q()), "dense0")
        _hip.check(lib.idqn_finish_step_factored(*args, _hip.FACTORED_REST, q()), "adam")
        return
    chk(hip.hipEventRecord(e_fork, q()), "record")
    chk(hip.hipStreamWaitEvent(sB, e_fork, 0), "wait")
    _hip.check(lib.idqn_backward_rest(agent._handle, q()), "rest")  # (host order only: the library wants it before the update)
    _hip.check(lib.idqn_finish_step_factored(*args, _hip.FACTORED_DENSE0, sB), "dense0")
    chk(hip.hipEventRecord(e_join, sB), "record")
    if mode == "within":
        chk(hip.hipStreamWaitEvent(q(), e_join, 0), "wait")
    _hip.check(lib.idqn_finish_step_factored(*args, _hip.FACTORED_REST, q()), "adam")


def run(name, fn, n=steps, warm=40):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    chk(hip.hipStreamSynchronize(sB), "sync")
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        chk(hip.hipStreamSynchronize(sB), "sync")
        ts.append((time.perf_counter() - t0) / n * 1e6)
    print(f"  {name:34s} {np.median(ts):7.1f} us/step   ({', '.join('%.1f' % t for t in ts)})", flush=True)
    return float(np.median(ts))


print(f"== n_b {n_b} pattern {pattern} IDQN_CUS {os.environ.get('IDQN_CUS', '256')}", flush=True)
run("fused (shipped step)", step_fused)
# the Dense_0 update alone on the second stream: how fast does it stream from n_b CUs?
agent._learn(batches[0], flags=_hip.F_STOP_BEFORE_DENSE0_WGRAD)
args = factors()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    chk(hip.hipEventRecord(t0e, sB), "rec")
    _hip.check(lib.idqn_finish_step_factored(*args, _hip.FACTORED_DENSE0, sB), "dense0")
    chk(hip.hipEventRecord(t1e, sB), "rec")
    chk(hip.hipEventSynchronize(t1e), "sync")
    ms = C.c_float()
    chk(hip.hipEventElapsedTime(C.byref(ms), t0e, t1e), "elapsed")
    best = min(best, ms.value * 1e3)
    agent._learn(batches[0], flags=_hip.F_STOP_BEFORE_DENSE0_WGRAD)  # (re-arm the split state)
    torch.cuda.synchronize()
_hip.check(lib.idqn_backward_rest(agent._handle, q()), "rest")
_hip.check(lib.idqn_finish_step_factored(*args, _hip.FACTORED_DENSE0 | _hip.FACTORED_REST, q()), "finish")
torch.cuda.synchronize()
print(f"  Dense_0 update alone on stream B: {best:.1f} us = {476.0 / best:.2f} TB/s", flush=True)
run("split, serial on one stream", lambda i: step_split(i, "serial"))
run("split, overlap within the step", lambda i: step_split(i, "within"))
torch.cuda.synchronize()
losses = agent._losses.cpu().numpy()
assert np.isfinite(losses).all(), losses
print("  losses", losses, flush=True)
