"""CPU checks of the i-IQN oracle (extension, parity unpinned: the reference has no quantile code, README.md:3,10).

* the hand-derived numpy backward agrees with torch autograd (two independent restatements);
* the committed probes under tests/golden/fp_path_iqn_small*.json are what the oracle computes today;
* properties of the loss the kernels rely on (target = constant -> zero loss at the target, asymmetric weights).
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_numpy_backward_matches_autograd():
    from oracle import iqn_ref as I
    from oracle import qnet_ref as Q
    from oracle import torch_ref as T

    obs, A, feats, K, B, N = (20, 20, 4), 5, [32, 32, 32, 128], 2, 8, 4
    p, pt = I.init_params(0, obs, A, feats, K), I.init_params(1, obs, A, feats, K)
    batch = Q.synthetic_batch(2, B, obs, A, "cnn")
    taus = I.synthetic_taus(3, K, N, B)
    for k in range(K):
        loss, g, _ = I.loss_and_grads(Q.head(p, k), Q.head(pt, k), batch, tuple(taus[k]), 0.99)
        loss_t, g_t = T.iqn_loss_and_grads(Q.head(p, k), Q.head(pt, k), batch, tuple(taus[k]), 0.99)
        assert abs(loss - loss_t) <= 1e-12 * abs(loss)
        for n in g:
            assert np.abs(g[n] - g_t[n]).max() <= 1e-10 * (np.abs(g_t[n]).max() + 1e-30), n


def test_committed_probes_are_current():
    from oracle import iqn_ref as I
    from oracle import make_golden as G
    from oracle import qnet_ref as Q

    for name in ("iqn_small", "iqn_small_ragged"):
        obs, A, feats, K, B, N = G.IQN_CASES[name]
        p, pt, batch, taus = G.iqn_case_inputs(name)
        rec = json.load(open(os.path.join(GOLDEN, f"fp_path_{name}.json")))
        for k in range(K):
            loss, g, aux = I.loss_and_grads(Q.head(p, k), Q.head(pt, k), batch, tuple(taus[k]), rec["hyper"]["gamma"])
            assert abs(loss - rec["losses"][k]) <= 1e-12
            for leaf, r in rec["leaves"].items():
                assert np.allclose(g[leaf].reshape(-1)[r["idx"]], np.asarray(r["grad"])[k], rtol=1e-10, atol=1e-14)


def test_quantile_huber_properties():
    from oracle import iqn_ref as I

    d = np.array([-3.0, -1.0, -0.5, 0.0, 0.5, 1.0, 3.0])
    assert np.allclose(I.huber(d), [2.5, 0.5, 0.125, 0.0, 0.125, 0.5, 2.5])
    assert np.allclose(I.huber_grad(d), [-1, -1, -0.5, 0, 0.5, 1, 1])
    # cos features: exact values at tau = 0.5 are cos(i pi / 2)
    c = I.cos_features(np.full((1, 1), 0.5, np.float32))[0, 0]
    assert np.allclose(c[:4], [0.0, -1.0, 0.0, 1.0], atol=1e-15)
