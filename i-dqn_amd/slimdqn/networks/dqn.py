"""DQN agent on the HIP path: the K = 1 case of the same kernels, without the leading head axis and
without shift / sync.  Mirrors the reference's ``slimdqn/networks/dqn.py:13-95``."""
from slimdqn import _hip
from slimdqn.networks._agent import DeviceAgent


class DQN(DeviceAgent):
    def __init__(self, key, observation_dim, n_actions, features: list, architecture_type: str, learning_rate: float,
                 gamma: float, update_horizon: int, update_to_data: int, target_update_frequency: int,
                 adam_eps: float = 1e-8):
        super().__init__(key, observation_dim, n_actions, 1, features, architecture_type, learning_rate, gamma,
                         update_horizon, adam_eps, stacked=False)
        self.update_to_data = update_to_data
        self.target_update_frequency = target_update_frequency

    @property
    def cumulated_loss(self) -> float:
        return float(self._cum[0].item())

    @cumulated_loss.setter
    def cumulated_loss(self, value) -> None:
        self._cum.fill_(float(value))

    def update_online_params(self, step: int, replay_buffer) -> None:
        if step % self.update_to_data == 0:
            self._sample_and_learn(replay_buffer)  # = learn_on_batch(.., replay_buffer.sample()), one C call where it can be

    def learn_on_batch(self, params, params_target, optimizer_state, batch_samples):
        """dqn.py:60-73 (in place; the state arguments must be this agent's own)."""
        assert params is self.params and params_target is self.target_params and optimizer_state is self.optimizer_state
        losses = self._learn(batch_samples)
        return self.params, self.optimizer_state, losses[0]

    def update_target_params(self, step: int):
        if step % self.target_update_frequency == 0:
            self._ensure_handle(32)
            # K = 1: idqn_target_update is exactly `target_params = params.copy()` (dqn.py:52)
            _hip.check(_hip.lib().idqn_target_update(self._handle, _hip.current_stream()), "idqn_target_update")
            logs = {"loss": self.cumulated_loss / (self.target_update_frequency / self.update_to_data)}
            self._cum.zero_()
            return True, logs
        return False, {}

    def q_values(self, params, state):
        assert params is self.params or params is self.target_params
        return self._q_values(0 if params is self.params else 1, 0, state)

    def best_action(self, params, state, **kwargs):
        """dqn.py:88-92."""
        assert params is self.params or params is self.target_params
        return self._best_action(0 if params is self.params else 1, 0, state)
