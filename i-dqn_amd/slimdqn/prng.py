"""Tiny host PRNG-key helper standing in for ``jax.random`` at the call sites of the experiments.

The reference threads ``jax.random`` keys through ``train`` / ``select_action`` / ``best_action``
(``experiments/base/dqn.py:31``, ``slimdqn/sample_collection/utils.py:10``, ``slimdqn/networks/idqn.py:128``).
Bit-parity with threefry is not part of the contract (parity is defined at the
sampler / loss level on identical inputs), only the call structure is: a key is an opaque value
that can be split and consumed.  Keys here are ``numpy.random.SeedSequence`` objects.
"""
import numpy as np


def PRNGKey(seed):
    if isinstance(seed, np.random.SeedSequence):
        return seed
    if isinstance(seed, (int, np.integer)):
        return np.random.SeedSequence(int(seed))
    return np.random.SeedSequence([int(x) for x in np.asarray(seed).reshape(-1)])


def split(key, num=2):
    return PRNGKey(key).spawn(num)


def generator(key):
    return np.random.default_rng(PRNGKey(key))


def randint(key, low, high):
    """Uniform integer in [low, high) drawn from ``key`` (deterministic per key, like jax)."""
    k = PRNGKey(key)
    return int(np.random.default_rng(np.random.SeedSequence(k.entropy, spawn_key=k.spawn_key + (0x5EED,))).integers(low, high))


def uniform(key):
    k = PRNGKey(key)
    return float(np.random.default_rng(np.random.SeedSequence(k.entropy, spawn_key=k.spawn_key + (0xF10A7,))).random())
