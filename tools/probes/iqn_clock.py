"""In-kernel clock of the i-IQN Dense_0 forward GEMM (diagnostic build path: IDQN_IQN_CLOCK=1 makes every workgroup stamp
s_memtime / s_memrealtime around its k loop).  ~2 s of back-to-back steps on random data first (MI355X_MICROARCH.md, DVFS
give-back item 6), then the median over workgroups of  d(s_memtime) / d(s_memrealtime) x 100 MHz."""
import os
os.environ.setdefault("IDQN_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "i-dqn_amd", "libidqn_hip_debug.so"))  # debug build (__graft_entry__.build_debug()): the stamps / switches used here
import ctypes as C, os, sys, time
os.environ["IDQN_IQN_CLOCK"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from collections import namedtuple
from slimdqn import _hip
from slimdqn.networks.iiqn import iIQN
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
rng = np.random.default_rng(0)
agent = iIQN(0, (84, 84, 4), 6, 5, [32, 64, 64, 512], "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4, n_quantiles=32)
b = Batch(torch.from_numpy(rng.integers(0, 256, (32, 84, 84, 4), dtype=np.uint8)).cuda(),
          torch.from_numpy(rng.integers(0, 6, 32).astype(np.int32)).cuda(),
          torch.from_numpy(rng.standard_normal(32).astype(np.float32)).cuda(),
          torch.from_numpy(rng.integers(0, 256, (32, 84, 84, 4), dtype=np.uint8)).cuda(),
          torch.from_numpy((rng.random(32) < 0.05).astype(np.uint8)).cuda())
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 2.5:
    for _ in range(50): agent._learn(b)
    torch.cuda.synchronize(); n += 50
p, nbytes = C.c_void_p(), C.c_int64()
_hip.check(_hip.lib().idqn_debug_buffer(agent._handle, b"iqn_clk", C.byref(p), C.byref(nbytes)), "iqn_clk")
out = torch.empty(nbytes.value // 8, dtype=torch.int64, device="cuda")
C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(out.data_ptr()), p, C.c_size_t(nbytes.value), 3)
raw = out.cpu().numpy()
c = raw[:1024].reshape(-1, 4)[:240]
w = raw[1024:1024 + 240 * 16].reshape(240, 8, 2).astype(np.float64)
dt, dr = (c[:, 2] - c[:, 0]).astype(np.float64), (c[:, 3] - c[:, 1]).astype(np.float64)
ok = dr > 0
ghz = dt[ok] / dr[ok] * 0.1
print(f"{n} steps; forward GEMM k loop: {np.median(dr[ok]) / 100:.1f} us per workgroup (median), in-kernel clock "
      f"median {np.median(ghz):.3f} GHz  p10 {np.percentile(ghz, 10):.3f}  p90 {np.percentile(ghz, 90):.3f}  ({ok.sum()} workgroups)")
nk = 242
print("per k-step, cycles: wait for own LDS writes  median over waves %.0f   in the barrier  median %.0f  (wave 0: %.0f, wave 7: %.0f; p90 %.0f)" % (
    np.median(w[:, :, 0]) / nk, np.median(w[:, :, 1]) / nk, np.median(w[:, 0, 1]) / nk, np.median(w[:, 7, 1]) / nk, np.percentile(w[:, :, 1], 90) / nk))
print("barrier wait per k-step by wave:", " ".join("%.0f" % (np.median(w[:, i, 1]) / nk) for i in range(8)))
us = dr[ok] / 100
print("k loop per workgroup, us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f;  first start .. last end of the k loops: %.1f us" % (
    us.min(), np.percentile(us, 10), np.median(us), np.percentile(us, 90), us.max(), (c[ok, 3].max() - c[ok, 1].min()) / 100))
e = raw[1024 + 4096:1024 + 4096 + 480].reshape(240, 2).astype(np.float64)
t0 = e[:, 0].min()
print("kernel timeline, us from the first workgroup's entry: entries %.1f .. %.1f;  k loops start %.1f .. %.1f, end %.1f .. %.1f;  stores done %.1f .. %.1f" % (
    0.0, (e[:, 0].max() - t0) / 100, (c[:, 1].min() - t0) / 100, (c[:, 1].max() - t0) / 100, (c[:, 3].min() - t0) / 100, (c[:, 3].max() - t0) / 100,
    (e[:, 1].min() - t0) / 100, (e[:, 1].max() - t0) / 100))
