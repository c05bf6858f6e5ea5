"""CPU: the integer / fp64 oracle against the reference's known answers and captured traces."""
import numpy as np
import pytest

import kat_int_path as kat
from oracle.replay_ref import ReplayRef, Transition
from oracle.samplers_ref import PrioritizedRef, UniformRef
from oracle.sumtree_ref import SumTreeRef


def test_sumtree_known_answers():
    kat.check_sumtree_kat(SumTreeRef)


@pytest.mark.parametrize("ci", range(7))
def test_sumtree_reference_traces(ci):
    z, meta = kat.load_sumtree_traces()
    kat.replay_sumtree_trace(SumTreeRef, z, meta, ci, check_every_op=(meta[ci]["capacity"] <= 3000))


def test_uniform_known_answers():
    kat.check_uniform_kat(UniformRef)


def test_prioritized_known_answers():
    kat.check_prioritized_kat(PrioritizedRef)


@pytest.mark.parametrize("ci", range(3))
def test_uniform_reference_traces(ci):
    z, meta = kat.load_sampler_traces()
    kat.replay_uniform_trace(UniformRef, z, meta, ci)


@pytest.mark.parametrize("pi", range(3))
def test_prioritized_reference_traces(pi):
    z, meta = kat.load_sampler_traces()
    kat.replay_prioritized_trace(PrioritizedRef, z, meta, pi)


def test_replay_known_answers():
    kat.check_replay_kat(ReplayRef, UniformRef, Transition)


def test_prioritized_zero_root_is_broken_like_the_reference():
    s = PrioritizedRef(0, 4)
    s.add(0, priority=0.0)
    with pytest.raises(AttributeError):  # samplers.py:106-108
        s.sample(1)
