"""Network descriptor: the role ``DQNNet`` plays in the reference (``slimdqn/networks/architectures/dqn.py:32-70``).

The reference's flax module both defines the architecture and executes it; here execution is the
HIP path (``csrc/cnn_kernels.h`` / ``csrc/fc_kernels.h``), so ``DQNNet`` only carries the
architecture, produces the parameter pytree layout and initialises parameters with the reference's
initialiser families: ``xavier_uniform`` for the cnn's convs and dense layers (``:40``),
``lecun_normal`` for ``fc`` (``:62``), zero biases.  ``impala`` (``:54-60``) is outside the HIP
path's scope and raises.
"""
from typing import Sequence

import numpy as np


class DQNNet:
    def __init__(self, features: Sequence[int], architecture_type: str, n_actions: int):
        if architecture_type not in ("cnn", "fc"):
            raise NotImplementedError(
                f"architecture_type={architecture_type!r}: only 'cnn' and 'fc' have HIP kernels (impala is out of scope)")
        self.features = [int(f) for f in features]
        self.architecture_type = architecture_type
        self.n_actions = int(n_actions)

    def init_leaf(self, rng: np.random.Generator, name: str, shape, n_heads: int) -> np.ndarray:
        if name.endswith("bias"):
            return np.zeros((n_heads,) + tuple(shape), np.float32)
        receptive = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in, fan_out = receptive * shape[-2], receptive * shape[-1]
        if self.architecture_type == "cnn":  # xavier_uniform
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            return rng.uniform(-lim, lim, size=(n_heads,) + tuple(shape)).astype(np.float32)
        std = np.sqrt(1.0 / fan_in) / 0.87962566103423978  # lecun_normal: truncated normal, variance 1/fan_in
        z = rng.standard_normal((n_heads,) + tuple(shape))
        bad = np.abs(z) > 2
        while bad.any():
            z[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(z) > 2
        return (z * std).astype(np.float32)
