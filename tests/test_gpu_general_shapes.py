"""Every cnn shape the reference's DQNNet accepts, not only the ones the MFMA kernels are built for: the general-shape HIP
kernels (csrc/gcnn_kernels.h) against the fp64 oracle -- per-head loss, every leaf gradient, Q-values, greedy actions.
The first case is the reference's own smoke-test configuration (tests/test_atari.py:24-28: `--features 2 3 1 15`,
batch 3); the others cover odd channel counts, several dense layers, more than 32 actions and non-square frames, and
the Nature-CNN shape itself forced onto the general kernels (IDQN_CNN_GENERAL=1) as a second implementation.
Reference: slimdqn/networks/architectures/dqn.py:39-53,65-70, slimdqn/networks/idqn.py:96-131.
"""
import os
import subprocess
import sys
from collections import namedtuple

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # obs, features, A, K, B
    ((84, 84, 4), [2, 3, 1, 15], 6, 1, 3),          # the reference's smoke test
    ((36, 28, 3), [16, 48, 24, 100, 60], 5, 2, 20),  # RGB-like input, two hidden dense layers, odd widths
    ((20, 20, 1), [8, 8, 8], 40, 3, 33),             # one channel, no hidden dense layer, 40 actions, B > 32
    ((24, 24, 4), [32, 64, 64, 96], 4, 2, 8),        # Nature widths but a dense width the MFMA kernels do not take
]


def _run_case(obs, feats, A, K, B, seed=0):
    from oracle import qnet_ref as Q
    from slimdqn import _hip
    from slimdqn.networks.idqn import iDQN

    p = Q.init_params(seed, "cnn", obs, A, feats, K)
    pt = Q.init_params(seed + 1, "cnn", obs, A, feats, K)
    rng = np.random.default_rng(seed + 2)
    for n in p:
        if n.endswith("bias"):
            p[n] = (0.05 * rng.standard_normal(p[n].shape)).astype(np.float32)
            pt[n] = (0.05 * rng.standard_normal(p[n].shape)).astype(np.float32)
    batch = Q.synthetic_batch(seed + 3, B, obs, A, "cnn")
    batch[4][0] = True
    agent = iDQN(0, obs, A, K, feats, "cnn", 1e-3, 0.99, 1, 1, 10**9, 10**9, adam_eps=1e-8)
    assert [n for n, _, _ in agent._leaves] == [n for n, _ in Q.leaf_shapes("cnn", obs, A, feats)]
    agent._load_flat(agent._online, p)
    agent._load_flat(agent._target, pt)
    losses = agent._learn(Batch(*batch), flags=_hip.F_GRADS_ONLY).cpu().numpy()
    G = agent._flat_grad()
    for k in range(K):
        loss, grads, aux = Q.loss_and_grads(Q.head(p, k), Q.head(pt, k), batch, "cnn", 0.99)
        assert abs(losses[k] - loss) <= 1e-5 * max(1.0, abs(loss)), (k, losses[k], loss)
        for leaf, g in grads.items():
            assert np.abs(G[leaf][k] - g).max() <= 2e-5 * (np.abs(g).max() + 1e-12), (leaf, np.abs(G[leaf][k] - g).max(), np.abs(g).max())
        q = agent.q_values(agent.params, batch[0][: min(B, 32)], k).cpu().numpy()
        assert np.abs(q - aux["q"][: min(B, 32)]).max() <= 1e-5 * max(1.0, np.abs(aux["q"]).max())
        qt = agent.q_values(agent.target_params, batch[3][:2], k).cpu().numpy()
        assert np.abs(qt - aux["q_next"][:2]).max() <= 1e-5 * max(1.0, np.abs(aux["q_next"]).max())
        act = int(agent._best_action(0, k, np.asarray(batch[0][1])))  # host state: the idqn_act_host route
        assert act == int(np.argmax(aux["q"][1]))
    # a full step (Adam) against the oracle's
    agent._learn(Batch(*batch))
    mu = {n: np.zeros_like(a, dtype=np.float64) for n, a in p.items()}
    p2, _, _, _, _ = Q.learn_on_batch(p, pt, mu, {n: v.copy() for n, v in mu.items()}, np.zeros(K, np.int64), batch, "cnn", 0.99, 1e-3, 1e-8)
    got = agent._flat(agent._online)
    for leaf in p2:
        big = np.abs(grads[leaf]) > 1e-3 * np.abs(grads[leaf]).max() if leaf in grads else None
        err = np.abs(got[leaf] - p2[leaf])
        assert err.max() <= 1e-3 * 1.01 + 1e-6, leaf  # never more than one Adam step (lr) away
        assert np.median(err) <= 2e-5, (leaf, np.median(err))
    assert (agent._count.cpu().numpy() == 1).all()


@pytest.mark.parametrize("obs,feats,A,K,B", CASES)
def test_general_shapes_against_oracle(obs, feats, A, K, B):
    _run_case(obs, feats, A, K, B)


def test_nature_shape_on_the_general_kernels():
    """The same step on two independent HIP implementations: the MFMA plane kernels and the general kernels agree with
    the oracle on the Nature-CNN shape (child process: the switch is read once)."""
    code = ("import sys, os; sys.path[:0] = [%r, %r, %r]\n"
            "import test_gpu_general_shapes as T\n"
            "T._run_case((84, 84, 4), [32, 64, 64, 512], 6, 2, 32)\nprint('OK')\n" % (ROOT, os.path.join(ROOT, "i-dqn_amd"), os.path.join(ROOT, "tests")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, IDQN_CNN_GENERAL="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]


def test_reference_smoke_flags_run_as_a_script(tmp_path):
    """tests/test_atari.py:15-59 of the reference, flag for flag (its `--features 2 3 1 15`, batch 3)."""
    root = os.path.join(ROOT, "i-dqn_amd")
    argv = ["-en", "_test_dqn_Pong", "-s", "1", "-dw", "-f", "2", "3", "1", "15", "-rbc", "100", "-bs", "3", "-n", "1", "-gamma", "0.99",
            "-lr", "1e-4", "-horizon", "10", "-ne", "1", "-ntspe", "10", "-utd", "3", "-tuf", "3", "-nis", "3", "-ee", "0.01", "-ed", "4",
            "-at", "cnn"]
    code = ("import sys; sys.path.insert(0, %r); from experiments.atari.dqn import run; run(%r, save_root=%r)" % (root, argv, str(tmp_path)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
