"""Tiny host PRNG-key helper standing in for ``jax.random`` at the call sites of the experiments.

The reference threads ``jax.random`` keys through ``train`` / ``select_action`` / ``best_action``
(``experiments/base/dqn.py:31``, ``slimdqn/sample_collection/utils.py:10``, ``slimdqn/networks/idqn.py:128``).
Bit-parity with threefry is not part of the contract (parity is defined at the
sampler / loss level on identical inputs), only the call structure is: a key is an opaque value
that can be split and consumed.

A key is a pair of 64-bit words; ``split`` and the draws are counter-mode hashes of it (SplitMix64 finaliser), so
every operation costs the same no matter how many splits led to the key.  (The first version used
``numpy.random.SeedSequence.spawn``: the spawn path of a key that is split once per environment step grows by one
entry per step and every operation on it hashes the whole path -- 0.9 ms per split after 1500 steps, 9 ms per
environment step after 5000.)
"""
import numpy as np

_M = (1 << 64) - 1


def _mix(x: int) -> int:
    """SplitMix64 finaliser: a bijective 64-bit hash."""
    x = (x + 0x9E3779B97F4A7C15) & _M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M
    return x ^ (x >> 31)


class Key(tuple):
    """Opaque PRNG key: two 64-bit words."""

    __slots__ = ()


def PRNGKey(seed) -> Key:
    if isinstance(seed, Key):
        return seed
    if isinstance(seed, (int, np.integer)):
        s = int(seed) & _M
        return Key((_mix(s), _mix(s ^ 0xA5A5A5A5A5A5A5A5)))
    words = [int(x) & _M for x in np.asarray(seed).reshape(-1)]
    a, b = 0x243F6A8885A308D3, 0x13198A2E03707344
    for w in words:
        a, b = _mix(a ^ w), _mix(b + w)
    return Key((a, b))


def _derive(key: Key, tag: int) -> int:
    a, b = PRNGKey(key)
    return _mix(a ^ _mix((b + tag) & _M))


def split(key, num=2):
    """``num`` independent child keys (deterministic per key, like ``jax.random.split``)."""
    return [Key((_derive(key, 2 * i + 1), _derive(key, 2 * i + 2))) for i in range(num)]


def generator(key):
    """A numpy Generator seeded from the key (parameter initialisation)."""
    a, b = PRNGKey(key)
    return np.random.default_rng([a & 0xFFFFFFFF, a >> 32, b & 0xFFFFFFFF, b >> 32])


def randint(key, low, high):
    """Uniform integer in [low, high) drawn from ``key`` (deterministic per key, like jax)."""
    span = int(high) - int(low)
    assert span > 0
    # 64 random bits onto the span by multiply-shift (bias < span / 2^64)
    return int(low) + ((_derive(key, 0x5EED) * span) >> 64)


def uniform(key):
    """Uniform float in [0, 1) with 53 random bits."""
    return (_derive(key, 0xF10A7) >> 11) * (1.0 / (1 << 53))
