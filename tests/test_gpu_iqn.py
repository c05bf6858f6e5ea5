"""GPU parity of the i-IQN heads (BASELINE config 3, a labelled extension) against the fp64 oracle's committed probes.

The reference has no quantile code (README.md:3,10 only names i-IQN): the oracle (oracle/iqn_ref.py) restates the
published algorithm and these goldens come from it -- PARITY UNPINNED, like the rest of the fp path.
Bars: per-head loss within 1e-5 (relative to max(1, |loss|): the quantile loss sums 32 terms per sample), quantile values
within 2e-5, every leaf gradient within 3e-5 of its largest entry, greedy target actions identical.
"""
import json
import os
from collections import namedtuple

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
Batch = namedtuple("Batch", "state action reward next_state is_terminal")


def _setup(name):
    from oracle import make_golden as G
    from slimdqn.networks.iiqn import iIQN

    obs, A, feats, K, B, N = G.IQN_CASES[name]
    p, pt, batch, taus = G.iqn_case_inputs(name)
    rec = json.load(open(os.path.join(GOLDEN, f"fp_path_{name}.json")))
    h = rec["hyper"]
    agent = iIQN(0, obs, A, K, feats, "cnn", h["lr"], h["gamma"], h["n"], 1, 10**9, 10**9, adam_eps=h["eps"], n_quantiles=N)
    agent._load_flat(agent._online, p)
    agent._load_flat(agent._target, pt)
    st, a, r, s2, term = batch
    return agent, Batch(st, a, r, s2, term), taus, rec, (obs, A, feats, K, B, N, p, pt)


@pytest.mark.parametrize("name", ["iqn_small", "iqn_small_ragged", "iqn_atari_k5"])
def test_iqn_step_against_golden(name):
    agent, batch, taus, rec, (obs, A, feats, K, B, N, p, pt) = _setup(name)
    losses = agent._learn(batch, taus=taus).cpu().numpy()
    want = np.asarray(rec["losses"])
    assert np.abs(losses - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), (losses, want)
    # quantile values of head 0, greedy target action
    dbg = agent._debug("iqn_dbg").cpu().numpy().reshape(K, 2 * N + 33, 32)[0]
    z_on, z_tg, q_sel, a_star = dbg[:N, :B], dbg[N : 2 * N, :B], dbg[2 * N : 2 * N + A, :B].T, dbg[2 * N + 32, :B]
    assert np.array_equal(a_star.astype(np.int64), np.asarray(rec["a_star_head0"]))
    for got, key in ((z_on, "z_online_head0"), (z_tg, "z_target_head0"), (q_sel, "q_select_head0")):
        w = np.asarray(rec[key])
        assert np.abs(got - w).max() <= 2e-5 * max(1.0, np.abs(w).max()), key
    # first step from zero Adam state: mu = (1 - b1) g, so every leaf's gradient is read off mu
    mu = agent._flat(agent._mu)
    par = agent._flat(agent._online)
    for leaf, r in rec["leaves"].items():
        idx = np.asarray(r["idx"])
        g = mu[leaf].reshape(K, -1)[:, idx] / (1.0 - 0.9)
        wg, scale = np.asarray(r["grad"]), np.asarray(r["grad_absmax"])[:, None]
        assert (np.abs(g - wg) <= 3e-5 * scale + 1e-12).all(), (leaf, np.abs(g - wg).max(), scale.max())
        # post-Adam parameters: one step moves a parameter by at most lr; the update direction is sign-like for tiny
        # gradients, so the bar is a fraction of lr wherever the gradient is not negligible
        wp = np.asarray(r["param"])
        big = np.abs(wg) > 1e-3 * scale
        assert (np.abs(par[leaf].reshape(K, -1)[:, idx] - wp)[big] <= 0.02 * rec["hyper"]["lr"] + 2e-7 * np.abs(wp[big])).all(), leaf
    assert (agent._count.cpu().numpy() == 1).all()
    assert np.allclose(agent.cumulated_losses, want, rtol=2e-6, atol=1e-5)


def test_iqn_acting_against_oracle():
    from oracle import iqn_ref as I
    from oracle import qnet_ref as Q

    agent, batch, taus, rec, (obs, A, feats, K, B, N, p, pt) = _setup("iqn_small")
    rng = np.random.default_rng(5)
    for which, params, arena in ((0, p, agent.params), (1, pt, agent.target_params)):
        for head in range(K):
            state = batch.state[head + 2 * which]
            tau = rng.random((N, 1)).astype(np.float32)
            q = agent.q_values(arena, state, head, taus=tau).cpu().numpy()[0]
            act, want = I.greedy_action(Q.head(params, head), state, tau[:, 0])
            assert np.abs(q - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
            assert int(q.argmax()) == act
    # a batch of states through the same entry point
    states = batch.state[:7]
    tau = rng.random((N, 7)).astype(np.float32)
    q = agent.q_values(agent.params, states, 1, taus=tau).cpu().numpy()
    for i in range(7):
        _, want = I.greedy_action(Q.head(p, 1), states[i], tau[:, i])
        assert np.abs(q[i] - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


def test_iqn_learns_a_known_target():
    """Sanity beyond parity: with terminal transitions and reward 1 every target quantile is 1, so every online quantile
    of the taken action must move towards 1 and the quantile loss must fall."""
    from slimdqn.networks.iiqn import iIQN

    rng = np.random.default_rng(0)
    obs, A, K, N, B = (20, 20, 4), 3, 2, 8, 32
    agent = iIQN(3, obs, A, K, [32, 32, 32, 256], "cnn", 1e-3, 0.99, 1, 1, 10**9, 10**9, adam_eps=1e-8, n_quantiles=N)
    s = rng.integers(0, 256, size=(B,) + obs, dtype=np.uint8)
    batch = Batch(s, rng.integers(0, A, size=B).astype(np.int32), np.ones(B, np.float32), s, np.ones(B, bool))
    first = agent._learn(batch).cpu().numpy().copy()
    for _ in range(60):
        last = agent._learn(batch).cpu().numpy()
    assert (last < 0.2 * first).all(), (first, last)


def test_iqn_step_is_reproducible_bit_for_bit():
    """The tiled GEMMs cut the sample range of the weight gradient in two and the embedding / dL/dh kernels deal the
    fractions to several workgroups: every such partial is added in a fixed order (no atomics), so two runs of the same
    steps from the same state end with identical bits -- at N = 16, where every GEMM kernel and every grouped kernel runs."""
    from slimdqn.networks.iiqn import iIQN

    rng = np.random.default_rng(3)
    obs, A, K, N, B = (20, 20, 4), 4, 2, 16, 32
    s = rng.integers(0, 256, size=(B,) + obs, dtype=np.uint8)
    s2 = rng.integers(0, 256, size=(B,) + obs, dtype=np.uint8)
    batch = Batch(s, rng.integers(0, A, size=B).astype(np.int32), rng.standard_normal(B).astype(np.float32), s2, rng.random(B) < 0.1)
    taus = [rng.random((K, 3, N, B)).astype(np.float32) * 0.98 + 0.01 for _ in range(3)]
    runs = []
    for _ in range(2):
        agent = iIQN(11, obs, A, K, [32, 64, 32, 256], "cnn", 2.5e-4, 0.99, 1, 1, 10**9, 10**9, adam_eps=1e-6, n_quantiles=N)
        losses = [agent._learn(batch, taus=t).cpu().numpy().copy() for t in taus]
        runs.append((losses, agent._flat(agent._online), agent._flat(agent._mu)))
    for la, lb in zip(runs[0][0], runs[1][0]):
        assert np.array_equal(la, lb)
    for leaf in runs[0][1]:
        assert np.array_equal(runs[0][1][leaf], runs[1][1][leaf]), leaf
        assert np.array_equal(runs[0][2][leaf], runs[1][2][leaf]), leaf


@pytest.mark.parametrize("K,N", [(5, 32), (2, 16), (3, 64)])
def test_iqn_update_in_the_gradient_launch_over_several_steps(K, N):
    """Atari shape (K = 5, N = 32 is BASELINE config 3; other head / fraction counts change the item mix and the planned order): the Adam update of Dense_0/kernel rides in the weight gradient's epilogue of the merged gradient
    launch (csrc/iqn_gemm.h, k_iqn_d0_bwd_adam) behind a gate that the data-gradient items of the same rows open and the last
    weight-gradient item re-arms.  Several steps in a row: finite losses (a gate that gave up poisons them), and two runs from the
    same state end with identical bits (one split: no cross-workgroup sums)."""
    from slimdqn.networks.iiqn import iIQN

    rng = np.random.default_rng(8)
    obs, A, B = (84, 84, 4), 6, 32
    s = rng.integers(0, 256, size=(B,) + obs, dtype=np.uint8)
    s2 = rng.integers(0, 256, size=(B,) + obs, dtype=np.uint8)
    batch = Batch(s, rng.integers(0, A, size=B).astype(np.int32), rng.standard_normal(B).astype(np.float32), s2, rng.random(B) < 0.1)
    taus = [rng.random((K, 3, N, B)).astype(np.float32) * 0.98 + 0.01 for _ in range(4)]
    runs = []
    for _ in range(2):
        agent = iIQN(5, obs, A, K, [32, 64, 64, 512], "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4, n_quantiles=N)
        losses = [agent._learn(batch, taus=t).cpu().numpy().copy() for t in taus]
        runs.append((losses, agent._flat(agent._online)["Dense_0/kernel"], agent._flat(agent._nu)["Dense_0/kernel"]))
    for la, lb in zip(runs[0][0], runs[1][0]):
        assert np.isfinite(la).all() and np.array_equal(la, lb)
    assert np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    assert np.abs(runs[0][2]).max() > 0  # (the second moments moved: the epilogue ran)


def test_iqn_trainer_entry_point(tmp_path):
    """experiments/atari/iiqn.py (extension): the reference's trainer loop -- collect, update_online, T-step copy / shift,
    D-step sync, logs, checkpoint -- with the quantile agent on the synthetic Atari environment."""
    import pickle

    from experiments.atari.iiqn import run

    argv = ["-en", "t", "-s", "2", "-ne", "1", "-ntspe", "70", "-nis", "40", "-rbc", "100", "-nn", "2", "-nq", "8", "-at", "cnn",
            "-tuf", "20", "-tsf", "5", "-f", "32", "64", "64", "256", "-horizon", "30", "-bs", "32"]
    p, agent = run(argv, save_root=str(tmp_path))
    logs = [r for r in p["wandb"].records if "loss" in r]
    assert logs and all(np.isfinite(r["loss"]) and f"networks/{agent.n_networks - 1}_loss" in r for r in logs)
    assert int(agent._count[0].item()) >= 30
    model = pickle.load(open(os.path.join(p["save_path"], "models", "2"), "rb"))
    leaves = model["params"]["params"]
    assert "Embed_0" in leaves and leaves["Embed_0"]["kernel"].shape[-2] == 64
    assert all(np.isfinite(v).all() for m in leaves.values() for v in m.values())
