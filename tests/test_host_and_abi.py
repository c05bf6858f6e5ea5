"""CPU (no GPU): host-side logic of the product package and the C-ABI symbol table.

The device classes need a GPU; what runs here is exactly the code that is host code in the reference
too (index maps, trajectory accumulator, PRNG-key shim) plus the contract that libidqn_hip.so loads
and exports every symbol include/idqn_hip.h declares.  No compute call is made.
"""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from slimdqn import _hip

    header = open(os.path.join(ROOT, "include", "idqn_hip.h")).read()
    declared = set(re.findall(r"^(?:const char\*|int)\s+(\w+)\s*\(", header, flags=re.M))
    assert declared == set(_hip.SYMBOLS), declared ^ set(_hip.SYMBOLS)
    lib = _hip.lib()  # raises if the .so is missing or lacks a symbol
    assert lib.idqn_abi_version() == 4
    for name in declared:
        assert hasattr(lib, name)
    # ... and nothing else: the dynamic symbol table of the .so, restricted to defined global functions that are not the
    # toolchain's own (underscore-prefixed), is exactly the declared set -- no undeclared experiment hooks ride along
    import subprocess

    nm = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llvm-nm")
    out = subprocess.run([nm if os.path.exists(nm) else "nm", "-D", "--defined-only", _hip.LIB_PATH], check=True,
                         capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in ("T", "t")}
    exported = {n for n in exported if not n.startswith("_")}
    assert exported == declared, sorted(exported ^ declared)


def test_layout_matches_the_reference_pytree():
    from oracle import qnet_ref as Q
    from slimdqn import _hip

    cfg = _hip.make_config("cnn", 5, 6, (84, 84, 4), [32, 64, 64, 512], 32, 6.25e-5, 1.5e-4, 0.99)
    leaves, stride = _hip.layout(cfg)
    want = Q.leaf_shapes("cnn", (84, 84, 4), 6, [32, 64, 64, 512])
    assert [(n, s) for n, _, s in leaves] == want
    assert stride % 64 == 0 and all(off % 64 == 0 for _, off, _ in leaves)
    assert sum(int(np.prod(s)) for _, _, s in leaves) == 4046502
    cfg = _hip.make_config("fc", 3, 4, (8, 1, 1), [100, 100], 32, 3e-4, 1e-8, 0.99)
    assert [(n, s) for n, _, s in _hip.layout(cfg)[0]] == Q.leaf_shapes("fc", 8, 4, [100, 100])


def test_unsupported_shapes_are_refused_loudly():
    from slimdqn import _hip

    # every cnn shape the reference's DQNNet takes has a layout (shapes outside the MFMA kernels run on the general-shape
    # kernels, csrc/gcnn_kernels.h), with the flax leaf names in creation order
    for feats, n_dense in (([32, 64, 60, 512], 2), ([32, 64, 64, 500], 2), ([32, 64, 64], 1), ([2, 3, 1, 15], 2), ([8, 8, 8, 30, 20], 3)):
        leaves, stride = _hip.layout(_hip.make_config("cnn", 5, 6, (84, 84, 4), feats, 32, 1e-4, 1e-8, 0.99))
        names = [n for n, _, _ in leaves]
        assert names[:6] == [f"Conv_{i}/{w}" for i in range(3) for w in ("kernel", "bias")]
        assert names[6:] == [f"Dense_{i}/{w}" for i in range(n_dense) for w in ("kernel", "bias")]
        assert leaves[-2][2][1] == 6 and stride % 64 == 0
    # what has no meaning is still refused: fewer than three convs, zero widths, the quantile heads on a general shape
    for feats, kw in (([32, 64], {}), ([32, 0, 64, 512], {}), ([32, 64, 60, 512], {"n_quantiles": 8})):
        with pytest.raises(_hip.HipExtensionError):
            _hip.layout(_hip.make_config("cnn", 5, 6, (84, 84, 4), feats, 32, 1e-4, 1e-8, 0.99, **kw))
    from slimdqn.networks.architectures.dqn import DQNNet

    with pytest.raises(NotImplementedError):
        DQNNet([32, 64, 64, 512], "impala", 6)


def test_agent_without_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from slimdqn import _hip
    from slimdqn.networks.idqn import iDQN

    with pytest.raises(_hip.HipExtensionError):
        iDQN(0, (84, 84, 4), 6, 5, [32, 64, 64, 512], "cnn", 1e-4, 0.99, 1, 1, 200, 10)


def test_index_map_swap_remove_matches_oracle():
    from oracle.samplers_ref import UniformRef
    from slimdqn.sample_collection.samplers import IndexMap

    rng = np.random.default_rng(0)
    m, ref, live, nxt = IndexMap(), UniformRef(0), [], 0
    for _ in range(2000):
        if rng.random() < 0.6 or len(live) < 2:
            m.add(nxt); ref.add(nxt); live.append(nxt); nxt += 1
        else:
            k = live.pop(int(rng.integers(len(live)))); m.remove(k); ref.remove(k)
        assert m.index_to_key == ref.index_to_key and m.key_to_index == ref.key_to_index
    with pytest.raises(AssertionError):
        m.remove(-1)


def test_trajectory_accumulator_matches_oracle():
    """The product's host accumulator against the oracle restatement of replay_buffer.py:103-200."""
    from oracle.replay_ref import ReplayRef, Transition
    from oracle.samplers_ref import UniformRef
    from slimdqn.sample_collection.replay_buffer import TrajectoryAccumulator, TransitionElement

    rng = np.random.default_rng(1)
    for stack, n, gamma in [(4, 1, 0.99), (1, 3, 1.0), (4, 5, 0.9), (2, 2, 0.5)]:
        acc = TrajectoryAccumulator(stack, n, gamma)
        ref = ReplayRef(UniformRef(0), 2, 10**6, stack_size=stack, update_horizon=n, gamma=gamma)
        for i in range(300):
            obs = rng.integers(0, 255, size=(3, 2)).astype(np.uint8)
            term, trunc = bool(rng.random() < 0.1), bool(rng.random() < 0.05)
            r, a = float(rng.normal()), int(rng.integers(4))
            got = list(acc.push(TransitionElement(obs, a, r, term, trunc)))
            want = ref.accumulate(Transition(obs, a, r, term, trunc))
            assert len(got) == len(want)
            for g, w in zip(got, want):
                np.testing.assert_array_equal(g.state, w.state)
                np.testing.assert_array_equal(g.next_state, w.next_state)
                assert (g.action, g.reward, g.is_terminal, g.episode_end) == (w.action, w.reward, w.is_terminal, w.episode_end)


def test_prng_shim_is_deterministic_and_splittable():
    from slimdqn import prng

    k = prng.PRNGKey(3)
    a, b = prng.split(k)
    assert prng.randint(a, 0, 1000) == prng.randint(prng.split(prng.PRNGKey(3))[0], 0, 1000)
    assert prng.randint(a, 0, 10**9) != prng.randint(b, 0, 10**9)
    assert 0.0 <= prng.uniform(a) < 1.0
    assert all(0 <= prng.randint(prng.PRNGKey(s), 0, 5) < 5 for s in range(50))
