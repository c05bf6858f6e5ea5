"""i-IQN agent on the HIP path -- BASELINE config 3, a LABELLED EXTENSION (parity unpinned).

The reference snapshot has no quantile code: its README names i-IQN and points at another repository
(``/root/reference/README.md:3,10``).  This class keeps the reference's iDQN protocol (constructor arguments, chain of
K heads, ``update_online_params`` / ``update_target_params`` / ``best_action``, ``slimdqn/networks/idqn.py:27-134``) and
swaps the head for an implicit quantile network (Dabney et al. 2018) -- the algorithm and its sources are written down
in ``oracle/iqn_ref.py``, which is also what the parity tests compare against.

Quantile fractions are drawn on the host from numpy's PCG64 (the sampler's generator family, ``samplers.py:17``) and
handed to the library as data: a step is a deterministic function of (parameters, batch, fractions).  They travel through
two pinned staging buffers used in turn (asynchronous copies: the host does not wait for the stream each step).
Acting: ``argmax_a mean_l Z(s, tau_l)[a]`` over ``n_quantiles`` fractions, for a uniformly drawn head -- head AND
fractions are functions of the ``key`` the trainer passes, like ``iDQN.best_action`` (``idqn.py:126-131``).
"""
import ctypes as C

import numpy as np
import torch

from slimdqn import _hip, prng
from slimdqn.networks.idqn import iDQN


class iIQN(iDQN):
    def __init__(self, key, observation_dim, n_actions, n_networks: int, features: list, architecture_type: str,
                 learning_rate: float, gamma: float, update_horizon: int, update_to_data: int,
                 target_update_frequency: int, target_sync_frequency: int, adam_eps: float = 1e-8,
                 n_quantiles: int = 32):
        assert architecture_type == "cnn", "the quantile heads are built on the cnn trunk"
        assert 1 <= n_quantiles <= 64
        self._n_quantiles = int(n_quantiles)  # read by DeviceAgent._config: the layout gains Embed_0/{kernel,bias}
        super().__init__(key, observation_dim, n_actions, n_networks, features, architecture_type, learning_rate, gamma,
                         update_horizon, update_to_data, target_update_frequency, target_sync_frequency, adam_eps)
        self._tau_rng = np.random.Generator(np.random.PCG64(prng.randint(key, 0, 2**31 - 1)))
        self._tau_dev = None
        self._tau_act = torch.zeros(self._n_quantiles * 32, dtype=torch.float32, device="cuda")
        self._pins = {}  # name -> [two pinned host tensors, their numpy views, the events behind their last copies, turn]

    def _upload(self, name, host: np.ndarray, dst: torch.Tensor):
        """host -> dst[: host.size] through one of two pinned buffers used in turn; a buffer is written again only after the
        event recorded behind its previous copy (the copy itself is asynchronous: no stream synchronisation per step)."""
        ent = self._pins.get(name)
        if ent is None or ent[0][0].numel() < host.size or ent[0][0].dtype != dst.dtype:
            pins = [torch.empty(max(host.size, 1), dtype=dst.dtype).pin_memory() for _ in range(2)]
            ent = self._pins[name] = [pins, [p.numpy() for p in pins], [None, None], 0]
        pins, views, events, turn = ent
        if events[turn] is not None:
            events[turn].synchronize()
        views[turn][: host.size] = host.reshape(-1)
        dst[: host.size].copy_(pins[turn][: host.size], non_blocking=True)
        ev = events[turn] or torch.cuda.Event()
        ev.record()
        events[turn] = ev
        ent[3] = turn ^ 1

    def sample_fractions(self, batch_size: int) -> np.ndarray:
        """tau [K][3][N][B] in (0, 1): online, action-selection and target fractions of every head."""
        return self._tau_rng.random((self._K, 3, self._n_quantiles, batch_size)).astype(np.float32)

    def _learn(self, batch, flags=0, mean_divisor=None, taus=None):
        assert mean_divisor is None, "the quantile heads have no sharded-minibatch mode"
        s, s2 = self._dev(batch.state, torch.uint8), self._dev(batch.next_state, torch.uint8)
        assert tuple(s.shape[1:]) == self._obs, f"state shape {tuple(s.shape)} vs observation_dim {self._obs}"
        B = int(s.shape[0])
        assert B <= 32, "the quantile heads take minibatches of at most 32 samples"
        a = self._dev(batch.action, torch.int32)
        r = self._dev(batch.reward, torch.float32)
        t = self._dev(batch.is_terminal, torch.uint8)
        if taus is None:
            taus = self.sample_fractions(B)
        taus = np.ascontiguousarray(taus, np.float32)
        assert taus.shape == (self._K, 3, self._n_quantiles, B), taus.shape
        if self._tau_dev is None or self._tau_dev.numel() != taus.size:
            self._tau_dev = torch.empty(taus.size, dtype=torch.float32, device="cuda")
        self._upload("tau", taus, self._tau_dev)
        self._ensure_handle(B)
        self._keep = (s, s2, a, r, t)
        _hip.check(_hip.lib().idqn_iqn_learn_on_batch(self._handle, _hip.ptr(s), _hip.ptr(s2), _hip.ptr(a), _hip.ptr(r),
                                                      _hip.ptr(t), _hip.ptr(self._tau_dev), B, int(flags),
                                                      _hip.current_stream()), "idqn_iqn_learn_on_batch")
        return self._losses

    def _iqn_q(self, which, head, state, taus=None, want_action=False, key=None):
        st = getattr(state, "tensor", state)
        E = int(np.prod(self._obs))
        if isinstance(st, torch.Tensor) and st.is_cuda:
            s = self._dev(st, torch.uint8)
        else:  # a host state (the trainer's acting path): pinned staging like iDQN._best_action, no pageable upload
            src = np.ascontiguousarray(np.asarray(st)).astype(np.uint8, copy=False)
            if not hasattr(self, "_state_dev") or self._state_dev.numel() < src.size:
                self._state_dev = torch.empty(max(src.size, E), dtype=torch.uint8, device="cuda")
            self._upload("state", src, self._state_dev)
            s = self._state_dev[: src.size]
        assert s.numel() % E == 0, f"state of {s.numel()} bytes vs observation_dim {self._obs}"
        n = s.numel() // E
        if taus is None:
            # acting is a function of the key (idqn.py:128 draws the head from it; the fractions come from a child of it)
            rng = prng.generator(prng.split(key, 2)[1]) if key is not None else self._tau_rng
            taus = rng.random((self._n_quantiles, n)).astype(np.float32)
        taus = np.ascontiguousarray(taus, np.float32)
        assert taus.shape == (self._n_quantiles, n), taus.shape
        self._upload("tau_act", taus, self._tau_act)
        self._ensure_handle(32)
        self._keep_q = s
        if not hasattr(self, "_action_out"):
            self._action_out = torch.zeros(32, dtype=torch.int32, device="cuda")
        _hip.check(_hip.lib().idqn_iqn_q_values(self._handle, int(which), int(head), _hip.ptr(s), n, _hip.ptr(self._tau_act),
                                                _hip.ptr(self._q_out), _hip.ptr(self._action_out) if want_action else None,
                                                _hip.current_stream()), "idqn_iqn_q_values")
        return self._q_out[:n]

    def q_values(self, params, state, idx_params: int, taus=None):
        """Mean over the fractions of Z(s, tau) of head ``idx_params``: device tensor [n, A]."""
        assert params is self.params or params is self.target_params
        return self._iqn_q(0 if params is self.params else 1, idx_params, state, taus)

    def best_action(self, params, state, key, taus=None):
        idx_params = prng.randint(key, 0, self.n_networks)
        assert params is self.params or params is self.target_params
        self._iqn_q(0 if params is self.params else 1, idx_params, state, taus, want_action=True, key=key)
        return self._action_out[0]
