"""Known-answer checks for the integer / fp64 path, written against an abstract implementation.

Each ``check_*`` function restates one test of the reference (file:line cited) and takes the
classes under test as arguments, so the SAME checks run against the CPU oracle (``-m "not gpu"``)
and against the HIP-backed ``slimdqn`` classes (``-m gpu``).  ``replay_trace`` functions replay
the traces that ``oracle/make_golden.py`` captured from the reference itself.
"""
import hashlib
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------------------------------------
# reference tests/test_sum_tree.py
# ---------------------------------------------------------------------------------------------
def check_sumtree_kat(SumTree):
    # :16-18 negative capacity
    with pytest.raises(AssertionError):
        SumTree(capacity=-1)
    tree = SumTree(capacity=100)
    # :20-22 negative value
    with pytest.raises(AssertionError):
        tree.set(0, -1)
    # :24-27 capacity 1
    t1 = SumTree(capacity=1)
    t1.set(0, 1.5)
    assert t1.root == 1.5
    # :29-37 leftmost branch carries the value
    tree.set(0, 1.0)
    assert tree.get(0) == 1.0
    nodes = np.asarray(tree._nodes)
    leaf = tree._first_leaf_offset
    while leaf > 0:
        leaf = leaf // 2
        assert nodes[leaf] == 1.0
    # :39-46 vectorised set/get with float32 values
    tree = SumTree(capacity=100)
    tree.set(np.array([1, 2], dtype=np.int32), np.array([3.0, 4.0], dtype=np.float32))
    assert tree.get(1) == 3.0 and tree.get(2) == 4.0 and tree.root == 7.0
    # :48-55 duplicates
    tree = SumTree(capacity=100)
    tree.set(np.array([1, 1, 1, 2, 2], dtype=np.int32), np.array([3.0, 3.0, 3.0, 4.0, 4.0], dtype=np.float32))
    assert tree.get(1) == 3.0 and tree.get(2) == 4.0 and tree.root == 7.0
    # :57-58 capacity
    assert np.asarray(tree._nodes).size >= 100
    # :60-62 empty tree query raises ValueError
    with pytest.raises(ValueError):
        SumTree(capacity=100).query(1.0)
    # :64-66
    tree = SumTree(capacity=100)
    tree.set(5, 1.0)
    assert tree.query(0.99) == 5
    # :68-87 four-leaf tree
    tree = SumTree(capacity=4)
    tree.set(np.array([0, 1, 2, 3], dtype=np.int32), np.array([0.5, 1.0, 0.5, 0.5], dtype=np.float32))
    assert tree.root == 2.5 and tree._depth == 3 and np.asarray(tree._nodes).size == 7
    np.testing.assert_array_equal(np.asarray(tree._nodes), [2.5, 1.5, 1.0, 0.5, 1.0, 0.5, 0.5])
    out = tree.query(np.array([1.5, 1.0]))
    np.testing.assert_array_equal(out, np.array([2, 1], np.int32))
    assert out.dtype == np.int32
    # :89-106 update then query
    tree.set(0, 0.25)
    assert tree.root == 2.25
    assert tree.query(0.249) == 0 and tree.query(0.5) == 1 and tree.query(1.25) == 2
    # :108-128 eight leaves, identity query with integer targets
    tree = SumTree(capacity=8)
    tree.set(np.arange(8, dtype=np.int32), np.ones((8,), dtype=np.float32))
    assert tree.root == 8.0 and tree._depth == 4 and np.asarray(tree._nodes).size == 15
    np.testing.assert_array_equal(tree.query(np.arange(8, dtype=np.int32)), np.arange(8, dtype=np.int32))
    # :130-136 max_recorded_priority
    tree = SumTree(capacity=100)
    tree.set(0, 0)
    assert tree.max_recorded_priority == 1
    for i in range(1, 32):
        tree.set(i, i)
        assert tree.max_recorded_priority == i
    # out-of-range targets (sum_tree.py:73-74): negative and == root
    with pytest.raises(ValueError):
        tree.query(np.array([-0.5]))
    with pytest.raises(ValueError):
        tree.query(np.array([float(tree.root)]))


def load_sumtree_traces():
    z = np.load(os.path.join(GOLDEN, "int_path_sumtree.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


def replay_sumtree_trace(SumTree, z, meta, ci, check_every_op=True):
    """Replays one captured trace; every node array must be BIT-identical to the reference's."""
    m = meta[ci]
    tree = SumTree(m["capacity"])
    assert tree._depth == m["depth"] and tree._first_leaf_offset == m["first_leaf"]
    for j in range(m["n_ops"]):
        idx, val = z[f"c{ci}_op{j}_idx"], z[f"c{ci}_op{j}_val"]
        if idx.ndim == 0:
            idx, val = int(idx), float(val)
        tree.set(idx, val)
        assert tree.root == z[f"c{ci}_roots"][j], (ci, j)
        assert tree.max_recorded_priority == z[f"c{ci}_maxp"][j], (ci, j)
        if check_every_op or j == m["n_ops"] - 1:
            got = hashlib.sha256(np.ascontiguousarray(np.asarray(tree._nodes, np.float64)).tobytes()).hexdigest()
            assert got == m["digests"][j], f"node array differs from the reference after op {j} of trace {ci}"
    nodes = np.asarray(tree._nodes)
    assert nodes.size == m["n_nodes"]
    if f"c{ci}_nodes" in z.files:
        np.testing.assert_array_equal(nodes, z[f"c{ci}_nodes"])
    np.testing.assert_array_equal(nodes[z[f"c{ci}_probe_idx"]], z[f"c{ci}_probe_val"])
    out = tree.query(z[f"c{ci}_q_targets"])
    assert out.dtype == np.int32
    np.testing.assert_array_equal(out, z[f"c{ci}_q_out"])


# ---------------------------------------------------------------------------------------------
# reference tests/test_samplers.py
# ---------------------------------------------------------------------------------------------
def check_prioritized_kat(Prioritized):
    s = Prioritized(seed=0, max_capacity=10)  # tests/test_samplers.py:14
    for key, prio in zip([0, 1, 2, 3, 4], [1.0, 2.0, 3.0, 4.0, 0.0]):
        s.add(key, priority=prio)
    out = s.sample(5)
    np.testing.assert_array_less(out, 4)  # :24-25 zero priority never sampled
    np.testing.assert_array_equal(out, [3, 1, 0, 0, 3])  # reference output, seed 0
    s.update(keys=np.array([2, 3]), priorities=np.array([0.0, 0.0]))
    np.testing.assert_array_less(s.sample(5), 2)  # :27-31
    s.remove(0)
    np.testing.assert_array_equal(s.sample(5), 1)  # :33-35
    assert list(s._index_to_key) == [4, 1, 2, 3]


def check_uniform_kat(Uniform):
    u = Uniform(0)
    for k in range(10):
        u.add(k)
    np.testing.assert_array_equal(u.sample(8), [8, 6, 5, 2, 3, 0, 0, 0])  # reference output
    with pytest.raises(AssertionError):
        u.remove(99)  # samplers.py:27
    with pytest.raises(AssertionError):
        Uniform(0).sample(1)  # samplers.py:41


def load_sampler_traces():
    z = np.load(os.path.join(GOLDEN, "int_path_samplers.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


def replay_uniform_trace(Uniform, z, meta, ci):
    u = Uniform(meta[ci]["seed"])
    got = []
    for op, arg in z[f"u{ci}_script"]:
        if op == 0:
            u.add(int(arg))
        elif op == 1:
            u.remove(int(arg))
        else:
            got.append(u.sample(int(arg)))
    np.testing.assert_array_equal(np.concatenate(got), z[f"u{ci}_samples"])
    np.testing.assert_array_equal(np.asarray(list(u._index_to_key)), z[f"u{ci}_final_index_to_key"])


def replay_prioritized_trace(Prioritized, z, meta, pi):
    m = [x for x in meta if x["kind"] == "prioritized"][pi]
    s = Prioritized(m["seed"], m["cap"], m["alpha"])
    for rec in m["recs"]:
        if rec["op"] == 0:
            s.add(rec["key"], priority=rec["prio"])
        elif rec["op"] == 1:
            s.remove(rec["key"])
        elif rec["op"] == 3:
            s.update(np.asarray(rec["keys"], np.int32), np.asarray(rec["prios"], np.float64))
        else:
            assert float(s._sum_tree.root) == rec["root"]
            np.testing.assert_array_equal(s.sample(rec["n"]), rec["out"])
    assert [int(k) for k in s._index_to_key] == m["final_index_to_key"]
    np.testing.assert_array_equal(np.asarray(s._sum_tree._nodes), z[f"p{pi}_final_nodes"])


# ---------------------------------------------------------------------------------------------
# reference tests/test_replay_buffer.py
# ---------------------------------------------------------------------------------------------
OBS = (84, 84)


def check_replay_kat(ReplayBuffer, Uniform, Transition):
    """tests/test_replay_buffer.py:51-299 (pack/unpack at :21-49 is the identity; no compression here)."""
    # ---- testAddUpToCapacity :51-88
    rb = ReplayBuffer(sampling_distribution=Uniform(seed=0), batch_size=32, max_capacity=10, stack_size=4,
                      update_horizon=1, gamma=1.0, compress=False)
    trs = []
    for i in range(16):
        trs.append(Transition(np.full(OBS, i), i, i, False, False))
        rb.add(trs[-1])
    mem = rb._memory
    assert len(mem) == 10 and list(mem.keys()) == list(range(5, 15))
    for i in range(5, 15):
        np.testing.assert_array_equal(
            mem[i].state, np.array([t.observation for t in trs[i - 3 : i + 1]]).transpose(1, 2, 0))
        np.testing.assert_array_equal(
            mem[i].next_state, np.array([t.observation for t in trs[i - 2 : i + 2]]).transpose(1, 2, 0))
        assert mem[i].action == trs[i].action and mem[i].reward == trs[i].reward
        assert mem[i].is_terminal == 0 and mem[i].episode_end == 0
    # ---- testNSteprewards :90-108
    rb = ReplayBuffer(sampling_distribution=Uniform(seed=0), batch_size=32, max_capacity=10, stack_size=4,
                      update_horizon=5, gamma=1.0, compress=False)
    for i in range(50):
        rb.add(Transition(np.full(OBS, i), 0, 2.0, False))
    for _ in range(10):
        np.testing.assert_array_equal(np.asarray(rb.sample().reward), np.ones(32) * 10.0)
    # ---- testGetStack :110-136
    rb = ReplayBuffer(sampling_distribution=Uniform(seed=0), batch_size=32, max_capacity=50, stack_size=4,
                      update_horizon=5, gamma=1.0, compress=False)
    for i in range(11):
        rb.add(Transition(np.full(OBS, i), 0, 0, False))
    mem = rb._memory
    for i in mem:
        assert tuple(mem[i].state.shape) == OBS + (4,)
    np.testing.assert_array_equal(np.zeros(OBS + (3,)), mem[0].state[:, :, :3])
    for i in range(4):
        np.testing.assert_array_equal(np.full(OBS, i), mem[3].state[:, :, i])
    # ---- testSampleTransitionBatch :138-183
    rb = ReplayBuffer(sampling_distribution=Uniform(seed=0), batch_size=2, max_capacity=10, stack_size=1,
                      update_horizon=1, gamma=0.99, compress=False)
    index_to_id = []
    for i in range(50):
        terminal = i % 4 == 0
        rb.add(Transition(np.full(OBS, i), 0, 0, terminal, False))
        if not terminal:
            index_to_id.append(i)
    i2k = list(rb._sampling_distribution._index_to_key)
    indices = np.random.default_rng(seed=0).integers(len(i2k), size=len(i2k))
    batch = rb.sample(size=len(indices))
    exp_s = np.array([np.full(OBS + (1,), index_to_id[i2k[i]]) for i in indices])
    exp_n = np.array([np.full(OBS + (1,), index_to_id[i2k[i]] + 1) for i in indices])
    exp_t = np.array([int(((index_to_id[i2k[i]] + 1) % 4) == 0) for i in indices])
    np.testing.assert_array_equal(np.asarray(batch.state), exp_s)
    np.testing.assert_array_equal(np.asarray(batch.next_state), exp_n)
    np.testing.assert_array_equal(np.asarray(batch.action), np.zeros(len(indices)))
    np.testing.assert_array_equal(np.asarray(batch.reward), np.zeros(len(indices)))
    np.testing.assert_array_equal(np.asarray(batch.is_terminal), exp_t)
    # ---- testSamplingWithTerminalInTrajectory :185-228
    rb = ReplayBuffer(sampling_distribution=Uniform(seed=0), batch_size=2, max_capacity=10, stack_size=1,
                      update_horizon=3, gamma=1.0, compress=False)
    for i in range(10):
        rb.add(Transition(np.full(OBS, i), action=i * 2, reward=i, is_terminal=i == 3, episode_end=False))
    indices = np.random.default_rng(seed=0).integers(rb.add_count, size=5)
    batch = rb.sample(size=5)
    exp_s = np.array([np.full(OBS + (1,), i) if i < 3 else np.full(OBS + (1,), i + 1) for i in indices])
    exp_a = np.array([i * 2 if i < 3 else (i + 1) * 2 for i in indices])
    exp_r = np.array([3, 6, 5, 15, 18, 21, 24])
    exp_t = np.array([1, 1, 1, 0, 0, 0, 0])
    np.testing.assert_array_equal(np.asarray(batch.state), exp_s)
    np.testing.assert_array_equal(np.asarray(batch.action), exp_a)
    np.testing.assert_array_equal(np.asarray(batch.reward), exp_r[indices])
    np.testing.assert_array_equal(np.asarray(batch.is_terminal), exp_t[indices])
    # ---- testKeyMappingsForSampling :230-299
    cap = 10
    rb = ReplayBuffer(sampling_distribution=Uniform(seed=0), batch_size=32, max_capacity=cap, stack_size=1,
                      update_horizon=1, gamma=0.99, compress=False)
    sampler = rb._sampling_distribution
    for i in range(cap + 1):
        rb.add(Transition(np.full(OBS, i), i, i, False, False))
    for i in range(cap):
        assert sampler._key_to_index[i] == i and sampler._index_to_key[i] == i
    rb.add(Transition(np.full(OBS, cap + 1), cap + 1, cap + 1, False, False))
    assert 0 not in sampler._key_to_index
    assert sampler._index_to_key[0] != 0
    assert cap in sampler._key_to_index
    assert sampler._index_to_key[sampler._key_to_index[cap]] == cap
    indices = np.random.default_rng(seed=0).integers(len(sampler._index_to_key), size=32)
    keys = [sampler._index_to_key[i] for i in indices]
    samples = rb.sample()
    for i, key in enumerate(keys):
        np.testing.assert_array_equal(np.asarray(samples.state)[i], np.full(OBS, key)[..., None])
        np.testing.assert_array_equal(np.asarray(samples.next_state)[i], np.full(OBS, key + 1)[..., None])
        assert samples.action[i] == key and samples.reward[i] == key
        assert samples.is_terminal[i] == 0 and samples.episode_end[i] == 0
    # empty buffer (replay_buffer.py:217)
    with pytest.raises(AssertionError):
        ReplayBuffer(sampling_distribution=Uniform(seed=0), batch_size=2, max_capacity=4).sample()
