"""GPU: the HBM sum tree / samplers / replay buffer (through the C ABI) are BIT-exact with the reference --
its own known-answer tests restated, plus the traces captured from the reference (tests/golden/int_path_*)."""
import numpy as np
import pytest

import kat_int_path as kat

pytestmark = pytest.mark.gpu


def _classes():
    from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
    from slimdqn.sample_collection.samplers import PrioritizedSamplingDistribution, UniformSamplingDistribution
    from slimdqn.sample_collection.sum_tree import SumTree

    return SumTree, UniformSamplingDistribution, PrioritizedSamplingDistribution, ReplayBuffer, TransitionElement


def test_sumtree_known_answers():
    kat.check_sumtree_kat(_classes()[0])


@pytest.mark.parametrize("ci", range(7))
def test_sumtree_reference_traces(ci):
    z, meta = kat.load_sumtree_traces()
    kat.replay_sumtree_trace(_classes()[0], z, meta, ci, check_every_op=(meta[ci]["capacity"] <= 3000))


def test_sumtree_large_set_is_chunked_in_order():
    """> 4096 updates in one call: host de-dup + ascending chunks must equal the oracle bit for bit."""
    from oracle.sumtree_ref import SumTreeRef

    rng = np.random.default_rng(11)
    idx = rng.integers(50000, size=9000).astype(np.int32)
    val = rng.random(9000) * 3
    a, b = _classes()[0](50000), SumTreeRef(50000)
    a.set(idx, val)
    b.set(idx, val)
    np.testing.assert_array_equal(a._nodes, b.nodes)


def test_uniform_known_answers():
    kat.check_uniform_kat(_classes()[1])


def test_prioritized_known_answers():
    kat.check_prioritized_kat(_classes()[2])


@pytest.mark.parametrize("ci", range(3))
def test_uniform_reference_traces(ci):
    """The reference-captured uniform add / remove / sample traces (u0..u2) on the PRODUCT sampler, not only the oracle."""
    z, meta = kat.load_sampler_traces()
    kat.replay_uniform_trace(_classes()[1], z, meta, ci)


@pytest.mark.parametrize("pi", range(3))
def test_prioritized_reference_traces(pi):
    z, meta = kat.load_sampler_traces()
    kat.replay_prioritized_trace(_classes()[2], z, meta, pi)


def test_replay_known_answers():
    SumTree, Uniform, Prioritized, ReplayBuffer, Transition = _classes()
    kat.check_replay_kat(ReplayBuffer, Uniform, Transition)


def test_replay_with_prioritized_sampler_matches_oracle():
    """BASELINE config 4's sampler wiring: add(priority=...) / update / sample through the HBM tree."""
    from oracle.replay_ref import ReplayRef, Transition as TRef
    from oracle.samplers_ref import PrioritizedRef

    SumTree, Uniform, Prioritized, ReplayBuffer, Transition = _classes()
    # tree capacity > replay capacity: add() inserts the new key before evicting the oldest (replay_buffer.py:209-213)
    a = ReplayBuffer(Prioritized(3, 100, 0.6), batch_size=16, max_capacity=64, stack_size=4, update_horizon=3, gamma=0.9)
    b = ReplayRef(PrioritizedRef(3, 100, 0.6), batch_size=16, max_capacity=64, stack_size=4, update_horizon=3, gamma=0.9)
    rng = np.random.default_rng(5)
    for i in range(150):
        obs = rng.integers(0, 256, size=(12, 12), dtype=np.uint8)
        term, pr, rew = bool(rng.random() < 0.07), float(rng.random() * 2), float(rng.normal())
        a.add(Transition(obs, int(i % 5), rew, term, False), priority=pr)
        b.add(TRef(obs, int(i % 5), rew, term, False), priority=pr)
    assert a.add_count == b.add_count
    for _ in range(5):
        x, y = a.sample(), b.sample()
        np.testing.assert_array_equal(np.asarray(x.state), y.state)
        np.testing.assert_array_equal(np.asarray(x.next_state), y.next_state)
        np.testing.assert_array_equal(np.asarray(x.action), y.action)
        np.testing.assert_array_equal(np.asarray(x.reward), y.reward.astype(np.float32))
        np.testing.assert_array_equal(np.asarray(x.is_terminal), y.is_terminal)
        keys = np.asarray(list(a._sampling_distribution._index_to_key))[:4].astype(np.int32)
        pr = np.array([0.5, 0.0, 2.5, 1.0])
        a.update(keys, priorities=pr)
        b.update(keys, priorities=pr)
    np.testing.assert_array_equal(a._sampling_distribution._sum_tree._nodes, b.sampler.tree.nodes)


@pytest.mark.parametrize("capacity", [1, 2, 5, 33, 1000, 1 << 16, (1 << 20) - 3])
def test_sumtree_query_paths_agree_with_the_oracle(capacity):
    """Minibatch-sized queries take the one-wave-per-query descent (up to 5 levels per memory round trip), larger
    ones the thread-per-query walk: both must give the oracle's leaves bit for bit, for every depth 1..21, and for
    targets on exact prefix-sum boundaries (where `t < left` flips)."""
    from oracle.sumtree_ref import SumTreeRef

    rng = np.random.default_rng(capacity)
    a, b = _classes()[0](capacity), SumTreeRef(capacity)
    for lo in range(0, capacity, 4096):
        idx = np.arange(lo, min(lo + 4096, capacity), dtype=np.int32)
        val = rng.integers(0, 4, idx.size).astype(np.float64) * 0.25  # exact sums, many zero-priority leaves
        if lo == 0:
            val[0] = 0.5
        a.set(idx, val)
        b.set(idx, val)
    np.testing.assert_array_equal(a._nodes, b.nodes)
    root = b.root
    for n in (1, 32, 2048, 5000):
        t = rng.uniform(0, root, n)
        t[: n // 2] = np.floor(t[: n // 2] * 4) / 4  # multiples of 0.25: exactly on leaf boundaries
        t = np.minimum(t, np.nextafter(root, 0))
        np.testing.assert_array_equal(np.asarray(a.query(t)), b.query(t))
    with pytest.raises(ValueError):
        a.query(np.asarray([root]))


def _stream(rng, i, shape, dtype, pattern):
    """One synthetic transition; `pattern` scripts (terminal, truncated) so that runs of very short truncated episodes
    (which make no elements at all) stretch the transitions an alive element's frames lie back."""
    if np.issubdtype(dtype, np.integer):
        obs = rng.integers(0, 200, size=shape).astype(dtype)
    else:
        obs = rng.normal(size=shape).astype(dtype)
    term, trunc = pattern(i, rng)
    return obs, int(rng.integers(0, 6)), float(rng.normal()), bool(term), bool(trunc)


@pytest.mark.parametrize("shape,dtype,stack,n,cap", [
    ((12, 12), np.uint8, 4, 3, 64),      # Atari-like: the packed uint8 x 4 path
    ((84, 84), np.uint8, 4, 1, 40),      # the real frame size
    ((8,), np.float32, 1, 1, 50),        # LunarLander-like: stack 1, f32 (generic path)
    ((5, 3), np.int64, 3, 5, 16),        # odd stack depth, 8-byte elements
    ((7, 9), np.uint8, 4, 2, 16),        # frame bytes not a multiple of 4: generic path for uint8 x 4
])
def test_frame_ring_buffer_equals_the_reference_elements(shape, dtype, stack, n, cap):
    """The frame-ring store (one HBM write per frame, stacks assembled by replay_gather_stacked) must hold exactly
    the elements the reference's accumulator materialises: zero frames before the episode start, early-terminal
    horizons, the terminal flush, truncated episodes, FIFO eviction -- for every alive key, bit for bit."""
    from oracle.replay_ref import ReplayRef, Transition as TRef
    from oracle.samplers_ref import UniformRef

    SumTree, Uniform, Prioritized, ReplayBuffer, Transition = _classes()
    a = ReplayBuffer(Uniform(1), batch_size=8, max_capacity=cap, stack_size=stack, update_horizon=n, gamma=0.97)
    b = ReplayRef(UniformRef(1), batch_size=8, max_capacity=cap, stack_size=stack, update_horizon=n, gamma=0.97)
    rng = np.random.default_rng(17)
    pattern = lambda i, r: (r.random() < 0.06, r.random() < 0.05)
    for i in range(5 * cap + 37):
        obs, act, rew, term, trunc = _stream(rng, i, shape, dtype, pattern)
        a.add(Transition(obs, act, rew, term, trunc))
        b.add(TRef(obs, act, rew, term, trunc))
        if i % 61 == 0 and a.add_count:
            x, y = a.sample(), b.sample()
            np.testing.assert_array_equal(np.asarray(x.state), y.state)
            np.testing.assert_array_equal(np.asarray(x.next_state), y.next_state)
            np.testing.assert_array_equal(np.asarray(x.reward), y.reward.astype(np.float32))
    assert a.add_count == b.add_count and list(a._memory.keys()) == list(b.memory.keys())
    for key in b.memory:
        got, want = a._memory[key], b.memory[key]
        np.testing.assert_array_equal(got.state, want.state)
        np.testing.assert_array_equal(got.next_state, want.next_state)
        assert got.state.dtype == want.state.dtype and got.state.shape == want.state.shape
        assert got.action == want.action and got.reward == want.reward
        assert got.is_terminal == want.is_terminal and got.episode_end == want.episode_end


@pytest.mark.parametrize("frame", [(6, 6), (8, 8)])  # 36-byte frames (byte copies) and 64-byte frames (16-byte copies)
def test_frame_ring_grows_when_elementless_transitions_pile_up(frame):
    """Episodes shorter than the horizon that are truncated make no elements, so alive elements can refer to frames
    arbitrarily many transitions back; the ring must grow rather than overwrite them."""
    from oracle.replay_ref import ReplayRef, Transition as TRef
    from oracle.samplers_ref import UniformRef

    SumTree, Uniform, Prioritized, ReplayBuffer, Transition = _classes()
    cap, stack, n = 8, 4, 3
    a = ReplayBuffer(Uniform(2), batch_size=4, max_capacity=cap, stack_size=stack, update_horizon=n, gamma=0.9)
    b = ReplayRef(UniformRef(2), batch_size=4, max_capacity=cap, stack_size=stack, update_horizon=n, gamma=0.9)
    rng = np.random.default_rng(3)

    def pattern(i, r):
        phase = i % 400
        if phase < 40:
            return (phase % 13 == 12, False)     # ordinary episodes: elements are made
        return (False, phase % 3 == 2)           # 120 three-step truncated episodes: no elements at all

    n0 = None
    for i in range(1300):
        obs, act, rew, term, trunc = _stream(rng, i, frame, np.uint8, pattern)
        a.add(Transition(obs, act, rew, term, trunc))
        b.add(TRef(obs, act, rew, term, trunc))
        n0 = n0 or a._n_frames
        if i % 97 == 0:
            for key in b.memory:
                np.testing.assert_array_equal(a._memory[key].state, b.memory[key].state)
                np.testing.assert_array_equal(a._memory[key].next_state, b.memory[key].next_state)
    assert a._n_frames > n0, "the scripted stream should have exhausted the initial ring"
    assert a.add_count == b.add_count
    x, y = a.sample(), b.sample()
    np.testing.assert_array_equal(np.asarray(x.state), y.state)
    np.testing.assert_array_equal(np.asarray(x.action), y.action)


@pytest.mark.parametrize("capacity,alpha", [(7, 1.0), (64, 0.6), (1000, 1.0), (1 << 17, 0.5)])
def test_prioritized_sampler_device_paths_match_the_oracle(capacity, alpha):
    """Round 4: add / remove / sample run as single launches with the index -> key map in HBM (no host read of the moved
    priority, one mailbox read per sample).  A long random add / remove / update / sample sequence on the product class
    against the oracle (itself pinned by the reference-captured traces): node arrays bit for bit after every operation on
    the small trees, sampled keys identical, the device map equal to the host mirror -- holes at both ends, hole == last,
    zero priorities, duplicate updates."""
    from oracle.samplers_ref import PrioritizedRef

    Prioritized = _classes()[2]
    a, b = Prioritized(9, capacity, alpha), PrioritizedRef(9, capacity, alpha)
    rng = np.random.default_rng(capacity)
    live, next_key, n_ops = [], 0, 400 if capacity <= 1000 else 150
    check_every = capacity <= 1000
    for op in range(n_ops):
        r = rng.random()
        if (r < 0.45 and len(live) < capacity) or len(live) < 2:
            pr = float(rng.choice([0.0, rng.random() * 3, rng.random() * 1e-3, 5.0]))
            a.add(next_key, priority=pr)
            b.add(next_key, priority=pr)
            live.append(next_key)
            next_key += int(rng.integers(1, 4))
        elif r < 0.70:
            k = live.pop(int(rng.choice([0, len(live) - 1, rng.integers(len(live))])))
            a.remove(k)
            b.remove(k)
        elif r < 0.85:
            ks = np.asarray(rng.choice(live, size=min(len(live), int(rng.integers(1, 6)))), dtype=np.int32)  # may repeat a key
            ps = np.where(rng.random(ks.size) < 0.2, 0.0, rng.random(ks.size) * 2)
            a.update(ks, ps)
            b.update(ks, ps)
        elif b.tree.root > 0.0:
            n = int(rng.integers(1, 40))
            np.testing.assert_array_equal(a.sample(n), b.sample(n))
        if check_every or op == n_ops - 1:
            np.testing.assert_array_equal(a._sum_tree._nodes, b.tree.nodes, err_msg=f"op {op}")
            assert list(a._index_to_key) == list(b.index_to_key)
            np.testing.assert_array_equal(a._i2k_dev[: len(live)].cpu().numpy(), np.asarray(b.index_to_key, np.int32))
    with pytest.raises(KeyError):  # the reference's first statement is `self._key_to_index[key]` (samplers.py:90)
        a.remove(10**8)


def test_prioritized_sampler_empty_tree_branch_consumes_the_stream_like_the_reference():
    """root == 0: the reference falls into `super().sample(size).keys` (AttributeError) AFTER drawing `integers` from the
    generator (samplers.py:106-108).  The product draws its uniforms before it learns the root, so it has to rewind and
    replay that draw: the generator states must be equal afterwards."""
    from oracle.samplers_ref import PrioritizedRef

    Prioritized = _classes()[2]
    a, b = Prioritized(4, 16, 1.0), PrioritizedRef(4, 16, 1.0)
    for k in range(3):
        a.add(k, priority=0.0)
        b.add(k, priority=0.0)
    for s in (a, b):
        with pytest.raises(AttributeError):
            s.sample(5)
    assert a._rng_key.bit_generator.state == b.rng.bit_generator.state
    a.update(np.asarray([1], np.int32), np.asarray([2.0]))
    b.update(np.asarray([1], np.int32), np.asarray([2.0]))
    np.testing.assert_array_equal(a.sample(8), b.sample(8))


def test_uniform_sampler_device_map_and_device_samples():
    """SURVEY 8b lists the samplers' index <-> key maps among the device state.  The uniform sampler keeps its map on the host
    by default (its output feeds host code); with `enable_device_map` every add / remove also writes the HBM copy
    (sampler_map_set) and `sample_device` maps the generator's indices on the device (sampler_map_indices): after a random
    add / remove sequence the device map equals the host list and twin samplers return the same keys either way.  Likewise
    the prioritized sampler's all-device `sample_device` against its host-returning `sample`."""
    _, Uniform, Prioritized, _, _ = _classes()
    a, b = Uniform(5), Uniform(5)
    a.enable_device_map(256)
    rng = np.random.default_rng(3)
    live, nxt = [], 0
    for _ in range(300):
        if len(live) < 2 or (rng.random() < 0.6 and len(live) < 256):
            a.add(nxt); b.add(nxt); live.append(nxt); nxt += 1
        else:
            k = live.pop(int(rng.integers(len(live))))
            a.remove(k); b.remove(k)
    np.testing.assert_array_equal(a._i2k_dev[: len(live)].cpu().numpy(), np.asarray(b._index_to_key, np.int32))
    for n in (1, 7, 64):
        np.testing.assert_array_equal(a.sample_device(n).cpu().numpy(), b.sample(n))
    a.enable_device_map(256)  # (re-enabling rebuilds the copy from the host list)
    np.testing.assert_array_equal(a._i2k_dev[: len(live)].cpu().numpy(), np.asarray(b._index_to_key, np.int32))
    pa, pb = Prioritized(8, 128, 0.7), Prioritized(8, 128, 0.7)
    for k in range(90):
        pr = float(rng.random() * 2)
        pa.add(k, priority=pr); pb.add(k, priority=pr)
    for k in (0, 89, 40, 41):
        pa.remove(k); pb.remove(k)
    for n in (1, 32, 100):
        np.testing.assert_array_equal(pa.sample_device(n).cpu().numpy(), pb.sample(n))


def test_prioritized_sample_past_the_live_entries_is_an_index_error():
    """A descent that ends on an empty leaf behind the live entries (drift in the node sums; here forced by writing mass
    into a leaf the map has no key for) is the reference's IndexError from `self._index_to_key[index]` (samplers.py:114) --
    not a stale key of a removed item read from the device map."""
    Prioritized = _classes()[2]
    a = Prioritized(3, 8, 1.0)
    for k in range(3):
        a.add(k, priority=0.0)
    a._sum_tree.set(np.asarray([5], np.int32), np.asarray([2.0]))  # leaf 5 carries all the mass, the map has 3 entries
    with pytest.raises(IndexError):
        a.sample(4)
    _, keys, _, status = a._sum_tree.query_host(np.asarray([0.5]), index_to_key=a._i2k_dev, n_live=3)
    assert status & 4 and keys[0] == -1
    leaves, _, _, status = a._sum_tree.query_host(np.asarray([0.5]))  # no map: leaves only, nothing to check
    assert leaves[0] == 5 and not status & 4


def test_uniform_device_map_refuses_adds_past_its_capacity():
    """enable_device_map(capacity) allocates `capacity` entries: an add past them must fail on the host, not write HBM."""
    Uniform = _classes()[1]
    u = Uniform(0)
    u.enable_device_map(4)
    for k in range(4):
        u.add(k)
    with pytest.raises(IndexError):
        u.add(4)
    assert len(u._index_to_key) == 4  # nothing was appended
    u.remove(1)
    u.add(9)
    np.testing.assert_array_equal(u._i2k_dev.cpu().numpy(), np.asarray(u._index_to_key, np.int32))


@pytest.mark.parametrize("n,obs_bytes", [(1, 16), (32, 28224), (77, 4100)])
def test_replay_gather_of_whole_pair_stores(n, obs_bytes):
    """replay_gather / replay_gather_scalars (stores that keep (state, next_state) pairs per slot; ReplayBuffer.sample's fetch
    + stack, replay_buffer.py:223-229): plain row gathers, bit-exact against numpy indexing, repeated slots included."""
    import ctypes as C

    import torch

    from slimdqn import _hip

    rng = np.random.default_rng(n)
    cap = 50
    store = torch.from_numpy(rng.integers(0, 256, (cap, 2, obs_bytes), dtype=np.uint8)).cuda()
    act = torch.from_numpy(rng.integers(0, 18, cap).astype(np.int32)).cuda()
    rew = torch.from_numpy(rng.standard_normal(cap).astype(np.float32)).cuda()
    term = torch.from_numpy((rng.random(cap) < 0.3).astype(np.uint8)).cuda()
    slots_h = rng.integers(0, cap, n).astype(np.int32)
    slots = torch.from_numpy(slots_h).cuda()
    s_out = torch.empty((n, obs_bytes), dtype=torch.uint8, device="cuda")
    s2_out = torch.empty_like(s_out)
    a_out = torch.empty(n, dtype=torch.int32, device="cuda")
    r_out = torch.empty(n, dtype=torch.float32, device="cuda")
    t_out = torch.empty(n, dtype=torch.uint8, device="cuda")
    lib, q = _hip.lib(), _hip.current_stream()
    _hip.check(lib.replay_gather(_hip.ptr(store), C.c_int64(obs_bytes), _hip.ptr(slots), n, _hip.ptr(s_out), _hip.ptr(s2_out), q), "replay_gather")
    _hip.check(lib.replay_gather_scalars(_hip.ptr(act), _hip.ptr(rew), _hip.ptr(term), _hip.ptr(slots), n, _hip.ptr(a_out), _hip.ptr(r_out),
                                         _hip.ptr(t_out), q), "replay_gather_scalars")
    torch.cuda.synchronize()
    host = store.cpu().numpy()
    np.testing.assert_array_equal(s_out.cpu().numpy(), host[slots_h, 0])
    np.testing.assert_array_equal(s2_out.cpu().numpy(), host[slots_h, 1])
    np.testing.assert_array_equal(a_out.cpu().numpy(), act.cpu().numpy()[slots_h])
    np.testing.assert_array_equal(r_out.cpu().numpy().view(np.uint32), rew.cpu().numpy()[slots_h].view(np.uint32))
    np.testing.assert_array_equal(t_out.cpu().numpy(), term.cpu().numpy()[slots_h])


@pytest.mark.parametrize("pi", range(3))
def test_prioritized_traces_with_a_gradient_step_in_flight(pi):
    """The prioritized sampler launches on the tree's own stream so that `sample` does not wait for the learner's step
    (samplers.py:105-116 behind idqn.py:65-72).  Pinned here: with gradient steps queued on the caller's stream all the
    time, the reference-captured add / update / remove / sample traces still come out key for key -- and a sample returns
    while that work is still running (the query did not queue behind it)."""
    import time

    import torch

    from collections import namedtuple

    from oracle import qnet_ref as Q
    from slimdqn.networks.idqn import iDQN

    obs, A, feats, K, B = (84, 84, 4), 6, [32, 64, 64, 512], 5, 32
    agent = iDQN(0, obs, A, K, feats, "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
    Batch = namedtuple("Batch", "state action reward next_state is_terminal")
    batch = Batch(*(torch.from_numpy(x).cuda() for x in Q.synthetic_batch(3, B, obs, A, "cnn")))
    Prioritized = _classes()[2]
    overlapped = [0, 0]

    class Busy(Prioritized):  # every sample is issued with ~10 gradient steps (~3 ms) queued in front of it on the caller's stream
        def sample(self, size):
            for _ in range(10):
                agent._learn(batch)
            ev = torch.cuda.Event()
            ev.record()
            t0 = time.perf_counter()
            keys = super().sample(size)
            overlapped[0] += int(not ev.query())  # the steps were still running when the keys arrived
            overlapped[1] += 1
            self.last_wait = time.perf_counter() - t0
            return keys

    z, meta = kat.load_sampler_traces()
    kat.replay_prioritized_trace(Busy, z, meta, pi)
    torch.cuda.synchronize()
    assert np.isfinite(agent._losses.cpu().numpy()).all()
    assert overlapped[1] > 0 and overlapped[0] >= 0.8 * overlapped[1], overlapped


@pytest.mark.parametrize("sampler_kind,B", [("uniform", 32), ("uniform", 96), ("prioritized", 32)])
def test_learn_on_replay_is_sample_then_learn(sampler_kind, B):
    """`update_online_params` (idqn.py:65-72) as ONE call on the frame ring (idqn_learn_on_replay: the stacked gather inside the
    step's staging launch, slots as kernel arguments) against its two halves -- `replay_buffer.sample()` then
    `learn_on_batch`: the same keys from the same generator, and parameters, optimizer state and losses equal BIT for bit
    after several steps, episode starts (zero frames) and ring wrap included."""
    import torch

    from slimdqn.networks.idqn import iDQN

    SumTree, Uniform, Prioritized, ReplayBuffer, Transition = _classes()
    obs, A, K, feats = (84, 84, 4), 6, 2, [32, 64, 64, 512]

    def make():
        sampler = Uniform(5) if sampler_kind == "uniform" else Prioritized(5, 4096, 0.7)
        rb = ReplayBuffer(sampler, batch_size=B, max_capacity=300, stack_size=4, update_horizon=2, gamma=0.99)
        rng = np.random.default_rng(9)
        for i in range(700):  # more transitions than the ring holds: slots and frames have wrapped
            tr = Transition(rng.integers(0, 256, (84, 84), dtype=np.uint8), int(rng.integers(A)), float(rng.normal()),
                            bool(i % 37 == 36), bool(i % 91 == 90))
            rb.add(tr, **({"priority": float(rng.random() + 0.1)} if sampler_kind == "prioritized" else {}))
        rb.reuse_sample_buffers = True
        return rb, iDQN(0, obs, A, K, feats, "cnn", 6.25e-5, 0.99, 2, 1, 10**9, 10**9, adam_eps=1.5e-4)

    rb_a, agent_a = make()
    rb_b, agent_b = make()
    agent_b.fuse_replay_sampling = False
    for step in range(4):
        agent_a.update_online_params(step, rb_a)
        agent_b.update_online_params(step, rb_b)
    torch.cuda.synchronize()
    assert agent_a.__dict__.get("_replay_fused_ok") is True, "the fused path did not run"
    for name in ("_online", "_mu", "_nu", "_losses", "_cum"):
        np.testing.assert_array_equal(getattr(agent_a, name).cpu().numpy(), getattr(agent_b, name).cpu().numpy(), err_msg=name)
    assert rb_a._sampling_distribution._rng_key.bit_generator.state == rb_b._sampling_distribution._rng_key.bit_generator.state
