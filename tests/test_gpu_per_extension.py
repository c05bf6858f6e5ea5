"""GPU: the prioritized-replay EXTENSION.  No reference behaviour exists for it (the reference's
sample() drops the keys and its loss has no importance weights), so these tests check the device path against the
oracle's weighted loss / TD errors and against numpy restatements of the formulas -- parity unpinned by construction.
"""
from collections import namedtuple

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Batch = namedtuple("Batch", "state action reward next_state is_terminal")
LOSS_ATOL = 1e-5


@pytest.mark.parametrize("arch,obs,feats,A,K,B", [
    ("cnn", (20, 20, 4), [32, 32, 32, 128], 5, 3, 32),
    ("cnn", (20, 20, 4), [32, 32, 32, 128], 5, 2, 45),   # two sample blocks, ragged
    ("fc", 8, [24, 16], 4, 3, 32),
])
def test_weighted_loss_td_errors_and_gradients_match_the_oracle(arch, obs, feats, A, K, B):
    import torch

    from oracle import qnet_ref as Q
    from slimdqn import _hip
    from slimdqn.networks.idqn import iDQN

    p = Q.init_params(0, arch, obs, A, feats, K)
    pt = Q.init_params(1, arch, obs, A, feats, K)
    batch = Q.synthetic_batch(2, B, obs, A, arch)
    w = np.random.default_rng(3).uniform(0.05, 1.0, B).astype(np.float32)
    agent = iDQN(0, obs, A, K, feats, arch, 1e-3, 0.99, 1, 1, 10**9, 10**9)
    agent._load_flat(agent._online, p)
    agent._load_flat(agent._target, pt)
    agent._ensure_handle(B)
    w_dev, td_dev = torch.from_numpy(w).cuda(), torch.zeros((K, B), dtype=torch.float32, device="cuda")
    _hip.check(_hip.lib().idqn_set_per_buffers(agent._handle, _hip.ptr(w_dev), _hip.ptr(td_dev)), "set")
    losses = agent._learn(Batch(*batch), flags=_hip.F_GRADS_ONLY).cpu().numpy()
    grads = agent._flat_grad()
    td = td_dev.cpu().numpy()
    for k in range(K):
        loss, g, aux = Q.loss_and_grads(Q.head(p, k), Q.head(pt, k), batch, arch, 0.99, weights=w)
        assert abs(losses[k] - loss) <= LOSS_ATOL
        np.testing.assert_allclose(td[k], np.abs(aux["td"]), rtol=2e-5, atol=2e-6)
        for name in g:
            scale = max(np.abs(g[name]).max(), 1e-12)
            assert np.abs(grads[name][k] - g[name]).max() <= 2e-5 * scale, (k, name)
    # weights removed again -> the reference's plain mean
    _hip.check(_hip.lib().idqn_set_per_buffers(agent._handle, None, None), "clear")
    losses = agent._learn(Batch(*batch), flags=_hip.F_GRADS_ONLY).cpu().numpy()
    for k in range(K):
        assert abs(losses[k] - Q.loss_and_grads(Q.head(p, k), Q.head(pt, k), batch, arch, 0.99)[0]) <= LOSS_ATOL


def test_per_formulas_against_numpy():
    import torch

    from oracle.sumtree_ref import SumTreeRef
    from slimdqn import _hip
    from slimdqn.sample_collection.sum_tree import SumTree

    lib, q = _hip.lib(), _hip.current_stream()
    rng = np.random.default_rng(0)
    cap, n = 1000, 64
    a, b = SumTree(cap), SumTreeRef(cap)
    pri = rng.random(cap) * (rng.random(cap) < 0.8)
    a.set(np.arange(cap, dtype=np.int32), pri)
    b.set(np.arange(cap, dtype=np.int32), pri)
    # sumtree_set_one == set of one element
    for idx, val in ((3, 0.75), (999, 0.0), (3, 0.125)):
        _hip.check(lib.sumtree_set_one(_hip.ptr(a._nodes_dev), a._depth, idx, val, None, q), "set_one")
        b.set(np.asarray([idx], np.int32), np.asarray([val]))
    np.testing.assert_array_equal(a._nodes, b.nodes)
    root = b.root
    # per_sample_leaves: plain and stratified targets, same leaves as the oracle's query
    for strat in (0, 1):
        u = rng.random(n)
        u_dev, leaves = torch.from_numpy(u).cuda(), torch.empty(n, dtype=torch.int32, device="cuda")
        _hip.check(lib.per_sample_leaves(_hip.ptr(a._nodes_dev), a._depth, _hip.ptr(u_dev), n, strat, _hip.ptr(leaves), q),
                   "per_sample_leaves")
        t = (np.arange(n) + u) / n * root if strat else u * root
        np.testing.assert_array_equal(leaves.cpu().numpy(), b.query(np.minimum(t, np.nextafter(root, 0))))
    # importance weights
    lv = leaves.cpu().numpy()
    wts = torch.empty(n, dtype=torch.float32, device="cuda")
    _hip.check(lib.per_importance_weights(_hip.ptr(a._nodes_dev), a._depth, _hip.ptr(leaves), n, cap, 0.4, _hip.ptr(wts), q), "w")
    want = (cap * b.nodes[b.first_leaf + lv] / root) ** -0.4
    np.testing.assert_allclose(wts.cpu().numpy(), want / want.max(), rtol=1e-6)
    np.testing.assert_array_equal(leaves.cpu().numpy(), lv)  # every leaf was inside [0, n_items): left alone
    # a leaf at or past the item count (a descent that rounding sent onto an empty leaf) is pulled back, in place
    stray = torch.tensor([5, cap - 1, 40], dtype=torch.int32, device="cuda")
    _hip.check(lib.per_importance_weights(_hip.ptr(a._nodes_dev), a._depth, _hip.ptr(stray), 3, 41, 0.4, _hip.ptr(wts), q), "w")
    assert stray.cpu().numpy().tolist() == [5, 40, 40]
    # priorities from |td|
    td = rng.random((5, n)).astype(np.float32)
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    mx = torch.ones(1, dtype=torch.float64, device="cuda")
    td_dev, running = torch.from_numpy(td).cuda(), 1.0
    for red, f in ((0, lambda x: x.mean(0)), (1, lambda x: x.max(0))):
        _hip.check(lib.per_priorities_from_td(_hip.ptr(td_dev), 5, n, red, 1e-3, 0.6, _hip.ptr(out), _hip.ptr(mx), q), "pri")
        want = (f(td.astype(np.float64)) + 1e-3) ** 0.6
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-12)
        running = max(running, float(out.cpu().numpy().max()))
        assert float(mx.item()) == running  # the running maximum (starts at 1.0 like max_recorded_priority)
    td50 = torch.from_numpy(td * 50).cuda()
    _hip.check(lib.per_priorities_from_td(_hip.ptr(td50), 5, n, 1, 0.0, 1.0, _hip.ptr(out), _hip.ptr(mx), q), "pri")
    assert float(mx.item()) == float((td * 50).astype(np.float64).max())


def test_prioritized_learner_runs_the_whole_loop_on_the_device():
    """End to end: FIFO overwrite keeps exactly the alive slots sampleable, a step changes the sampled leaves'
    priorities to (mean_k |td| + eps)^alpha, and elements with larger TD error are drawn more often."""
    import torch

    from slimdqn.networks.idqn import iDQN
    from slimdqn.sample_collection.per import PrioritizedLearner, SlotPrioritizedSampler
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement

    cap, B, K = 64, 32, 3
    sampler = SlotPrioritizedSampler(0, cap, priority_exponent=0.6)
    rb = ReplayBuffer(sampler, batch_size=B, max_capacity=cap, stack_size=4, update_horizon=1, gamma=0.99)
    agent = iDQN(0, (20, 20, 4), 5, K, [32, 32, 32, 128], "cnn", 1e-3, 0.99, 1, 1, 10**9, 10**9)
    rng = np.random.default_rng(1)
    for i in range(150):
        rb.add(TransitionElement(rng.integers(0, 256, (20, 20), dtype=np.uint8), int(rng.integers(5)),
                                 float(10.0 if i % 7 == 0 else 0.1 * rng.normal()), bool(i % 50 == 49), False))
    assert len(sampler) == cap
    tree = sampler._sum_tree
    leaves0 = tree._nodes[tree._first_leaf_offset : tree._first_leaf_offset + cap]
    assert (leaves0 == 1.0).all()  # every newcomer entered with the running maximum (1.0 so far)
    learner = PrioritizedLearner(agent, rb, beta=0.5, eps=1e-3, reduce="mean")
    losses = learner.step()
    torch.cuda.synchronize()
    assert np.isfinite(losses.cpu().numpy()).all()
    lv = learner._leaves.cpu().numpy()
    td = learner._td_abs.cpu().numpy().astype(np.float64)
    want = (td.mean(0) + 1e-3) ** 0.6
    got = tree._nodes[tree._first_leaf_offset + lv]
    first = {}
    for pos, leaf in enumerate(lv):  # duplicates: the first occurrence wins (SumTree.set semantics)
        first.setdefault(int(leaf), pos)
    for leaf, pos in first.items():
        # leaf += (value - leaf), as np.add.at does: equal to the value up to one rounding
        assert abs(got[list(lv).index(leaf)] - want[pos]) <= 1e-12 * want[pos]
    assert abs(float(sampler._max_priority_dev.item()) - max(1.0, want.max())) <= 1e-12 * max(1.0, want.max())
    assert abs(tree._nodes[0] - tree._nodes[tree._first_leaf_offset : tree._first_leaf_offset + cap].sum()) < 1e-9
    for _ in range(60):
        learner.step()
    torch.cuda.synchronize()
    pri = tree._nodes[tree._first_leaf_offset : tree._first_leaf_offset + cap]
    big = np.asarray([abs(float(rb._reward64[s])) > 5 for s in range(cap)])
    assert pri[big].mean() > pri[~big].mean()  # the +10 rewards are the large-TD elements
    keys = sampler.sample(16)
    assert all(k in rb._memory for k in keys)
