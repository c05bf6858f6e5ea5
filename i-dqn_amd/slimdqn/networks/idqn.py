"""iDQN agent on the HIP path -- same constructor, methods and attributes as the reference's
``slimdqn/networks/idqn.py:27-134`` so that ``experiments/*/idqn.py`` and ``experiments/base/dqn.py``
drive it unchanged.

What differs is ownership: the reference rebinds immutable jax pytrees; here ``params`` /
``target_params`` / ``optimizer_state`` are views into HBM arenas that ``libidqn_hip.so`` updates
in place, so the T-step copy (``idqn.py:78``) is a real device copy ordered before the shift
(``:80``), and ``cumulated_losses`` (``:72``) is accumulated in f64 on the device and read back only
when a T-step needs the logs -- no per-step host sync.
"""
import numpy as np

from slimdqn import _hip, prng
from slimdqn.networks._agent import DeviceAgent


class iDQN(DeviceAgent):
    def __init__(self, key, observation_dim, n_actions, n_networks: int, features: list, architecture_type: str,
                 learning_rate: float, gamma: float, update_horizon: int, update_to_data: int,
                 target_update_frequency: int, target_sync_frequency: int, adam_eps: float = 1e-8, _local_heads=None):
        self.n_networks = n_networks
        # _local_heads = (first, count): hold only that window of the n_networks heads (head-parallel mode)
        first, count = _local_heads if _local_heads is not None else (0, n_networks)
        super().__init__(key, observation_dim, n_actions, count, features, architecture_type, learning_rate,
                         gamma, update_horizon, adam_eps, stacked=True, init_heads=(n_networks, first))
        self.update_to_data = update_to_data
        self.target_update_frequency = target_update_frequency
        self.target_sync_frequency = target_sync_frequency

    # ``cumulated_losses += losses`` (idqn.py:72) lives on the device; reading it synchronises
    @property
    def cumulated_losses(self) -> np.ndarray:
        return self._cum.cpu().numpy()

    @cumulated_losses.setter
    def cumulated_losses(self, value) -> None:
        import torch

        self._cum.copy_(torch.as_tensor(np.asarray(value, np.float64)))

    def update_online_params(self, step: int, replay_buffer) -> None:
        if step % self.update_to_data == 0:
            self._sample_and_learn(replay_buffer)  # = learn_on_batch(.., replay_buffer.sample()), one C call where it can be

    def learn_on_batch(self, params, params_target, optimizer_state, batch_samples):
        """idqn.py:96-109.  The three state arguments must be this agent's own (in-place update)."""
        assert params is self.params and params_target is self.target_params and optimizer_state is self.optimizer_state, \
            "the HIP path updates the agent's own arenas in place"
        losses = self._learn(batch_samples)
        return self.params, self.optimizer_state, losses

    def update_target_params(self, step: int):
        if step % self.target_update_frequency == 0:
            # target <- online (real copy), then window shift: idqn.py:78-80
            self._target_update()
            cum = self._all_cumulated_losses()
            denom = self.target_update_frequency / self.update_to_data
            logs = {"loss": np.mean(cum) / denom}
            for idx_network in range(self.n_networks):
                logs[f"networks/{idx_network}_loss"] = cum[idx_network] / denom
            self._cum.zero_()
            return True, logs
        if step % self.target_sync_frequency == 0:  # skipped on T-steps by the early return (idqn.py:89-92)
            self._target_sync()
        return False, {}

    # the three hooks the head-parallel agent (slimdqn/networks/head_parallel.py) overrides
    def _target_update(self) -> None:
        self._local_target_update()

    def _target_sync(self) -> None:
        self._local_target_sync()

    def _all_cumulated_losses(self) -> np.ndarray:
        return self.cumulated_losses

    def q_values(self, params, state, idx_params: int):
        """``network.apply(params[idx_params], state)`` (idqn.py:131): device tensor [n, A]."""
        assert params is self.params or params is self.target_params
        return self._q_values(0 if params is self.params else 1, idx_params, state)

    def best_action(self, params, state, key):
        """Greedy action of a uniformly drawn head; the head comes from the SAME key (idqn.py:126-131)."""
        idx_params = prng.randint(key, 0, self.n_networks)
        assert params is self.params or params is self.target_params
        return self._best_action(0 if params is self.params else 1, idx_params, state)
