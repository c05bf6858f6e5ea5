"""CPU: the fp oracle -- two independent restatements must agree, the reference's formula tests hold,
and the committed goldens are reproducible.  (Not reference-captured: parity unpinned, oracle/__init__.py.)"""
import json
import os

import numpy as np
import pytest

from oracle import make_golden as G
from oracle import qnet_ref as Q
from oracle import torch_ref as T

CASES = [("cnn", (20, 20, 4), [32, 32, 32, 128], 5, 2, 8), ("fc", 8, [50, 40], 4, 3, 16),
         ("cnn", (84, 84, 4), [32, 64, 64, 512], 6, 1, 3)]


def _setup(arch, obs, feats, A, K, B, seed=0):
    p = Q.init_params(seed, arch, obs, A, feats, K, np.float64)
    pt = Q.init_params(seed + 1, arch, obs, A, feats, K, np.float64)
    rng = np.random.default_rng(seed + 2)
    for n in p:
        if n.endswith("bias"):
            p[n] = p[n] + 0.05 * rng.standard_normal(p[n].shape)
    batch = list(Q.synthetic_batch(seed + 3, B, obs, A, arch))
    batch[4][0] = True
    return p, pt, tuple(batch)


@pytest.mark.parametrize("arch,obs,feats,A,K,B", CASES)
def test_numpy_and_autograd_restatements_agree(arch, obs, feats, A, K, B):
    p, pt, batch = _setup(arch, obs, feats, A, K, B)
    for k in range(K):
        l1, g1, _ = Q.loss_and_grads(Q.head(p, k), Q.head(pt, k), batch, arch, 0.99)
        l2, g2 = T.loss_and_grads(Q.head(p, k), Q.head(pt, k), batch, arch, 0.99)
        assert abs(l1 - l2) <= 1e-12 * max(1.0, abs(l2))
        for n in g1:
            assert np.abs(g1[n] - g2[n]).max() <= 1e-9 * (np.abs(g2[n]).max() + 1e-30), n


@pytest.mark.parametrize("arch,obs,feats,A,K,B", CASES[:2])
def test_gradients_match_central_differences(arch, obs, feats, A, K, B):
    """A third angle on the fp oracle that needs no automatic differentiation at all: the analytic gradients of
    `loss_and_grads` (the hand-written backward the HIP kernels are pinned to) against central differences of ITS OWN fp64
    loss in random coordinates of every leaf.  A backward pass that agreed with torch's autograd only because both shared a
    misreading of the forward (padding, flatten order, the target's stop-gradient) would still have to be the derivative of
    the loss the forward computes."""
    p, pt, batch = _setup(arch, obs, feats, A, K, B)
    rng = np.random.default_rng(17)
    k = 0
    hp = {n: v.astype(np.float64) for n, v in Q.head(p, k).items()}
    ht = {n: v.astype(np.float64) for n, v in Q.head(pt, k).items()}
    _, g, _ = Q.loss_and_grads(hp, ht, batch, arch, 0.99)
    for n, leaf in hp.items():
        flat = leaf.reshape(-1)
        for i in rng.choice(flat.size, size=min(6, flat.size), replace=False):
            old, eps = flat[i], 1e-5 * max(1.0, abs(flat[i]))
            flat[i] = old + eps
            lp = Q.loss_and_grads(hp, ht, batch, arch, 0.99)[0]
            flat[i] = old - eps
            lm = Q.loss_and_grads(hp, ht, batch, arch, 0.99)[0]
            flat[i] = old
            num, ana = (lp - lm) / (2 * eps), g[n].reshape(-1)[i]
            # (ReLU kinks: a unit that flips inside +-eps breaks the difference quotient, not the gradient -- allow 1e-4 relative
            # of the leaf's largest gradient, far above fp64 noise and far below any structural error)
            assert abs(num - ana) <= 1e-4 * (np.abs(g[n]).max() + 1e-12) + 1e-9, (n, int(i), num, ana)


def test_same_padding_geometry():
    # architectures/dqn.py:43-51 with flax's default padding="SAME": 84 -> 21 -> 11 -> 11, flatten 7744
    assert Q.same_pad(84, 8, 4) == (21, 2, 2)
    assert Q.same_pad(21, 4, 2) == (11, 1, 2)
    assert Q.same_pad(11, 3, 1) == (11, 1, 1)
    shapes = dict(Q.leaf_shapes("cnn", (84, 84, 4), 6, [32, 64, 64, 512]))
    assert shapes["Dense_0/kernel"] == (7744, 512)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 4046502  # params per head


def test_compute_target_and_loss_formulae():
    """tests/test_idqn.py:44-71 / tests/test_dqn.py:39-59 restated: target = r + (1-term) * gamma * max Q(s'),
    loss = (target - Q(s)[a])^2, for a single sample, with the same parameters online and target."""
    arch, obs, feats, A, K, B = CASES[0]
    p, _, batch = _setup(arch, obs, feats, A, K, 1, seed=5)
    for term in (False, True):
        s, a, r, s2, _ = batch
        b1 = (s, a, r, s2, np.array([term]))
        hp = Q.head(p, 1)
        loss, _, aux = Q.loss_and_grads(hp, hp, b1, arch, 0.94)
        q_next = Q.forward(hp, s2, arch)
        assert q_next.shape == (1, A)
        target = r[0] + (1 - int(term)) * 0.94 * q_next.max()
        assert aux["target"][0] == pytest.approx(target, rel=1e-14)
        pred = Q.forward(hp, s, arch)[0, a[0]]
        assert loss == pytest.approx((target - pred) ** 2, rel=1e-12)


def test_adam_first_step_is_sign_like_and_count_advances():
    theta, g = np.array([1.0, -2.0, 0.5]), np.array([0.3, -0.2, 0.0])
    new, m, v = Q.adam_update(theta, g, np.zeros(3), np.zeros(3), 0, lr=1e-3, eps=1e-8)
    np.testing.assert_allclose(m, 0.1 * g)
    np.testing.assert_allclose(v, 0.001 * g * g)
    np.testing.assert_allclose(new, theta - 1e-3 * np.sign(g), atol=1e-9)


def test_shift_and_sync_semantics():
    # idqn.py:13-24: shift: params[k] <- params[k+1]; sync: target[k] <- params[k-1] for k >= 1
    p = {"w": np.arange(4.0)[:, None] * np.ones((4, 3))}
    t = {"w": -np.ones((4, 3))}
    np.testing.assert_array_equal(Q.shift_params(p)["w"][:, 0], [1, 2, 3, 3])
    np.testing.assert_array_equal(Q.sync_target_params(p, t)["w"][:, 0], [-1, 0, 1, 2])


@pytest.mark.parametrize("name", ["cnn_small", "fc_lunar_k3"])
def test_goldens_are_reproducible(name):
    """The committed fp goldens are exactly what oracle/make_golden.py --fp computes."""
    arch, obs, A, feats, K, B, steps = G.FP_CASES[name]
    rec = json.load(open(os.path.join(G.GOLDEN, f"fp_path_{name}.json")))
    p, pt, batches = G.fp_case_inputs(name)
    p64 = {n: a.astype(np.float64) for n, a in p.items()}
    mu = {n: np.zeros_like(a) for n, a in p64.items()}
    nu = {n: np.zeros_like(a) for n, a in p64.items()}
    count = np.zeros(K, np.int64)
    h = rec["hyper"]
    for s, batch in enumerate(batches):
        p64, mu, nu, count, losses = Q.learn_on_batch(p64, pt, mu, nu, count, batch, arch, h["gamma"] ** h["n"],
                                                       h["lr"], h["eps"])
        np.testing.assert_allclose(losses, rec["steps"][s]["losses"], rtol=1e-12)
        for leaf, d in rec["steps"][s]["leaves"].items():
            np.testing.assert_allclose(p64[leaf].reshape(K, -1)[:, d["idx"]], d["param"], rtol=1e-12)


def test_batched_fp32_cpu_baseline_matches_fp64_oracle():
    """The timed CPU baseline (torch fp32, K heads batched) computes the same step as the oracle."""
    arch, obs, feats, A, K, B = "cnn", (84, 84, 4), [32, 64, 64, 512], 6, 2, 4
    p, pt, batch = _setup(arch, obs, feats, A, K, B, seed=9)
    step = T.BatchedStep(p, pt, A, 0.99, 6.25e-5, 1.5e-4)
    losses32 = step.step(batch)
    zeros = {n: np.zeros_like(a) for n, a in p.items()}
    new_p, _, _, _, losses64 = Q.learn_on_batch(p, pt, zeros, zeros, np.zeros(K, np.int64), batch, arch, 0.99,
                                                6.25e-5, 1.5e-4)
    np.testing.assert_allclose(losses32, losses64, atol=1e-5)
    for n in p:
        np.testing.assert_allclose(step.p[n].detach().numpy(), new_p[n], atol=2e-6)


def test_adam_restatement_matches_torch_optim_adam_over_many_steps():
    """optax.adam(lr, b1=.9, b2=.999, eps, eps_root=0) and torch.optim.Adam(lr, betas, eps) are the same recurrence
    (bias-corrected moments, eps outside the square root): an independent implementation must reproduce the oracle's
    `adam_update` step for step."""
    import torch

    rng = np.random.default_rng(0)
    theta0 = rng.standard_normal(50)
    grads = rng.standard_normal((25, 50)) * np.exp(rng.uniform(-6, 2, (25, 1)))
    lr, eps = 6.25e-5, 1.5e-4
    th, m, v = theta0.copy(), np.zeros(50), np.zeros(50)
    p = torch.nn.Parameter(torch.from_numpy(theta0.copy()))
    opt = torch.optim.Adam([p], lr=lr, betas=(0.9, 0.999), eps=eps)
    for t, g in enumerate(grads):
        th, m, v = Q.adam_update(th, g, m, v, t, lr, eps)
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        np.testing.assert_allclose(th, p.detach().numpy(), rtol=1e-12, atol=1e-15)


def test_impala_forward_and_the_reference_unit_tests_formulae():
    """The architecture the reference's OWN unit tests instantiate (tests/test_idqn.py:27-33: "impala", four features
    drawn from 1..9, 84x84x4 observations, 2..9 actions, gamma 0.94).  The oracle restates its forward pass
    (architectures/dqn.py:7-29,54-60: Conv 3x3 SAME, max_pool 3x3 / 2 SAME with -inf padding, two residual blocks per
    Stack) in numpy and in torch; the two must agree, and the three formulae those tests check -- compute_target,
    loss, best_action (tests/test_idqn.py:44-84) -- are evaluated on it.  (Forward only: the HIP path has no impala
    kernels -- SURVEY section 2 puts them out of scope -- and DQNNet refuses the architecture loudly.)"""
    import torch

    from oracle import torch_ref as T

    rng = np.random.default_rng(11)
    for _ in range(3):
        feats = [int(x) for x in rng.integers(1, 10, size=4)]
        A, K = int(rng.integers(2, 10)), int(rng.integers(1, 4))
        obs = (84, 84, 4)
        p = Q.init_params(int(rng.integers(1000)), "impala", obs, A, feats, K)
        names = [n for n, _ in Q.leaf_shapes("impala", obs, A, feats)]
        assert names[:10] == [f"Stack_0/Conv_{i}/{w}" for i in range(5) for w in ("kernel", "bias")] and names[-4:] == [
            "Dense_0/kernel", "Dense_0/bias", "Dense_1/kernel", "Dense_1/bias"]
        assert p["Dense_0/kernel"].shape == (K, 11 * 11 * feats[2], feats[3])  # 84 -> 42 -> 21 -> 11 under three SAME pools
        s = rng.random((1,) + obs)  # tests/utils.py:16-33 feeds float states in [0, 1)
        s2 = rng.random((1,) + obs)
        k = int(rng.integers(K))
        hp = Q.head(p, k)
        q, q2 = Q.forward(hp, s, "impala"), Q.forward(hp, s2, "impala")
        qt = T.forward_head({n: torch.as_tensor(a).double() for n, a in hp.items()}, torch.as_tensor(s), "impala").numpy()
        assert q.shape == (1, A) and np.abs(q - qt).max() <= 1e-12 * max(1.0, np.abs(q).max())
        for term in (0, 1):
            reward, action = float(rng.normal()), int(rng.integers(A))
            target = reward + (1 - term) * 0.94 * q2.max()                       # compute_target, idqn.py:120-124
            assert Q.td_target(q2, np.array([reward]), np.array([term]), 0.94)[0] == pytest.approx(target, rel=1e-15)
            loss = (q[0, action] - target) ** 2                                  # loss, idqn.py:114-118
            assert loss == pytest.approx(np.square(target - q[0, action]), rel=1e-15)
        assert int(np.argmax(q[0])) == int(q[0].argmax())                        # best_action, idqn.py:131 (first maximum)


def test_max_pool_same_matches_flax_semantics():
    x = np.arange(25, dtype=np.float64).reshape(1, 5, 5, 1) - 30.0  # all negative: a zero pad would win, -inf must not
    y = Q.max_pool_same(x)
    assert y.shape == (1, 3, 3, 1)
    np.testing.assert_array_equal(y[0, :, :, 0], np.array([[6, 8, 9], [16, 18, 19], [21, 23, 24]]) - 30.0)
