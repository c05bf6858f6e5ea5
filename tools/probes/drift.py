"""Free-running drift of the HIP step against the fp64 oracle (and of the oracle's own fp32 mode), per step."""
import os, sys
import numpy as np
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "i-dqn_amd")]
from collections import namedtuple
from oracle import make_golden as G
from oracle import qnet_ref as Q
from slimdqn.networks.idqn import iDQN

name, N = "cnn_small", int(os.environ.get("N", "100"))
arch, obs, A, feats, K, B, _ = G.FP_CASES[name]
p, pt, _ = G.fp_case_inputs(name)
h = G.FP_HYPER
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
agents = {}
for mode in ("bf16x3", "f32"):
    os.environ["IDQN_CONV"] = mode
    a = iDQN(0, obs, A, K, feats, arch, h["lr"], h["gamma"], h["n"], 1, 10**9, 10**9, adam_eps=h["eps"])
    a._load_flat(a._online, p); a._load_flat(a._target, pt)
    a._learn(Batch(*Q.synthetic_batch(1, B, obs, A, arch)), flags=1)  # creates the handle under this mode (grads only)
    agents[mode] = a
def fresh(dt):
    P = {n: x.astype(dt) for n, x in p.items()}
    return P, {n: np.zeros_like(x) for n, x in P.items()}, {n: np.zeros_like(x) for n, x in P.items()}, np.zeros(K, np.int64)
s64, s32 = fresh(np.float64), fresh(np.float32)
g_n = h["gamma"] ** h["n"]
err = {m: [] for m in ("bf16x3", "f32", "numpy-fp32")}
for s in range(N):
    batch = Q.synthetic_batch(1000 + s, B, obs, A, arch)
    P, mu, nu, cnt, want = Q.learn_on_batch(*s64, batch, arch, g_n, h["lr"], h["eps"], np.float64)[:5] if False else (None,)*5
    out = Q.learn_on_batch(s64[0], pt, s64[1], s64[2], s64[3], batch, arch, g_n, h["lr"], h["eps"], np.float64)
    s64 = out[:4]; want = out[4]
    out = Q.learn_on_batch(s32[0], {n: x.astype(np.float32) for n, x in pt.items()}, s32[1], s32[2], s32[3], batch, arch, g_n, h["lr"], h["eps"], np.float32)
    s32 = out[:4]; err["numpy-fp32"].append(float(np.abs(out[4] - want).max()))
    for m, a in agents.items():
        err[m].append(float(np.abs(a._learn(Batch(*batch)).cpu().numpy() - want).max()))
for m, e in err.items():
    e = np.asarray(e)
    print(f"{m:11s} max {e.max():.2e}  steps over 1e-5: {(e > 1e-5).sum()}  median {np.median(e):.2e}  first 25 max {e[:25].max():.2e}  first 50 max {e[:50].max():.2e}")
