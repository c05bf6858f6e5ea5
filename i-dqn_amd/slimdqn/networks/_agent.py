"""Device state shared by ``iDQN`` and ``DQN``: parameter arenas in HBM + the C-ABI handle.

All arithmetic of the gradient step runs in ``libidqn_hip.so``; torch tensors are storage.
Arenas are ``[K][head_stride]`` float32; ``params`` / ``target_params`` / ``optimizer_state`` are
pytrees of VIEWS into them with the flax leaf names of the reference
(``{"params": {"Conv_0": {"kernel", "bias"}, ...}}``, leading K axis for iDQN -- idqn.py:48-50).
"""
import ctypes as C

import os

import numpy as np
import torch

from slimdqn import _hip, prng
from slimdqn.networks.architectures.dqn import DQNNet


def _obs_triplet(observation_dim, arch):
    if isinstance(observation_dim, (int, np.integer)):
        return (int(observation_dim), 1, 1)
    dims = tuple(int(d) for d in observation_dim)
    if arch == "cnn":
        assert len(dims) == 3, "cnn expects observation_dim = (H, W, C)"
        return dims
    return (int(np.prod(dims)), 1, 1)


class _HostAction(int):
    """The greedy action as a host integer that still answers ``.item()`` like the device scalar the reference returns
    (``select_action`` calls ``.item()`` on it, slimdqn/sample_collection/utils.py:21)."""

    def item(self):
        return int(self)


class _PendingHostAction:
    """A greedy action whose launch is in flight (``idqn_act_host_begin``): ``.item()`` waits for it.  The trainer's
    ``collect_single_sample`` runs the replay bookkeeping of the previous transition between the two."""

    __slots__ = ("_agent", "_value")

    def __init__(self, agent):
        self._agent, self._value = agent, None

    def item(self):
        if self._value is None:
            a = self._agent
            try:
                _hip.check(_hip.lib().idqn_act_host_end(a._handle, C.c_void_p(a._act_out.data_ptr()), _hip.current_stream()),
                           "idqn_act_host_end")
                self._value = int(a._act_out_np[0])
            finally:  # the library has dropped its pending launch either way: a failure must not wedge every later call
                a._act_in_flight = None
        return self._value

    __int__ = __index__ = item


class DeviceAgent:
    lazy_host_actions = False  # True (the trainer of this build sets it): best_action on a host state returns a pending action

    def __init__(self, key, observation_dim, n_actions, n_heads, features, architecture_type, learning_rate, gamma,
                 update_horizon, adam_eps, stacked, init_heads=None):
        self.network = DQNNet(features, architecture_type, n_actions)
        self._K = int(n_heads)
        self._stacked = stacked  # iDQN leaves carry a leading K axis, DQN leaves do not
        self._arch = architecture_type
        self._obs = _obs_triplet(observation_dim, architecture_type)
        self._lr, self._eps = float(learning_rate), float(adam_eps)
        self.gamma, self.update_horizon = gamma, update_horizon
        self._gamma_n = float(gamma) ** update_horizon  # a Python double, folded like idqn.py:122
        _hip.lib()  # fail loudly (HipExtensionError) before any allocation if the extension is missing
        if not torch.cuda.is_available():
            raise _hip.HipExtensionError("no HIP device visible: the i-DQN step has no CPU path")
        self._leaves, self._P = _hip.layout(self._config(32))
        K, P = self._K, self._P
        dev = "cuda"
        self._online = torch.zeros((K, P), dtype=torch.float32, device=dev)
        self._target = torch.zeros((K, P), dtype=torch.float32, device=dev)
        self._mu = torch.zeros((K, P), dtype=torch.float32, device=dev)
        self._nu = torch.zeros((K, P), dtype=torch.float32, device=dev)
        # gradient arena (include/idqn_hip.h): [K][gP] small leaves | 64 reserved floats (the K losses live there, so
        # that one collective covers them) | [K][w0n] Dense_0/kernel gradients of the cnn
        names = [n for n, _, _ in self._leaves]
        self._w0 = (0, 0)
        if architecture_type == "cnn":
            i = names.index("Dense_0/kernel")
            self._w0 = (self._leaves[i][1], self._leaves[i + 1][1])
        self._gP = P - (self._w0[1] - self._w0[0])
        self._grad = torch.zeros(K * P + 64, dtype=torch.float32, device=dev)
        self._grad_small = self._grad[: K * self._gP + 64]
        self._grad_w0 = self._grad[K * self._gP + 64 :]
        self._losses = self._grad[K * self._gP : K * self._gP + K]
        self._losses_in_grad = True
        self._count = torch.zeros(K, dtype=torch.int32, device=dev)
        self._cum = torch.zeros(K, dtype=torch.float64, device=dev)
        self._handle, self._handle_batch = None, 0
        self._q_out = torch.zeros((32, n_actions), dtype=torch.float32, device=dev)
        # initial parameters (idqn.py:48-50 / dqn.py:29); target starts equal to online (idqn.py:56)
        rng = prng.generator(key)
        # init_heads = (K_global, first): this agent holds heads [first, first + K) of a K_global-head i-DQN and must
        # start from exactly the parameters the single-device agent gives those heads (head-parallel mode)
        k_all, first = init_heads if init_heads is not None else (K, 0)
        host = np.zeros((K, P), np.float32)
        for name, off, shape in self._leaves:
            leaf = self.network.init_leaf(rng, name, shape, k_all).reshape(k_all, -1)
            host[:, off : off + int(np.prod(shape))] = leaf[first : first + K]
        self._online.copy_(torch.from_numpy(host))
        self._target.copy_(self._online)
        self.params = self._tree(self._online)
        self.target_params = self._tree(self._target)
        self.optimizer_state = {"mu": self._tree(self._mu), "nu": self._tree(self._nu), "count": self._count}

    # ---- plumbing ----------------------------------------------------------------------------------
    def _config(self, max_batch):
        return _hip.make_config(self._arch, self._K, self.network.n_actions, self._obs, self.network.features,
                                max_batch, self._lr, self._eps, self._gamma_n,
                                n_quantiles=getattr(self, "_n_quantiles", 0))  # > 0: i-IQN heads (slimdqn/networks/iiqn.py)

    def _tree(self, arena):
        tree = {}
        for name, off, shape in self._leaves:
            mod, leaf = name.split("/")
            view = arena[:, off : off + int(np.prod(shape))].view((self._K,) + tuple(shape))
            tree.setdefault(mod, {})[leaf] = view if self._stacked else view[0]
        return {"params": tree}

    def _ensure_handle(self, batch):
        if self._handle is not None and batch <= self._handle_batch:
            return
        self._destroy_handle()
        cfg = self._config(max(batch, 32))
        h = C.c_void_p()
        _hip.check(_hip.lib().idqn_create(C.byref(cfg), _hip.ptr(self._online), _hip.ptr(self._target),
                                          _hip.ptr(self._mu), _hip.ptr(self._nu), _hip.ptr(self._grad),
                                          _hip.ptr(self._count), _hip.ptr(self._losses), _hip.ptr(self._cum),
                                          C.byref(h)), "idqn_create")
        self._handle, self._handle_batch = h, max(batch, 32)

    def _destroy_handle(self):
        pending = getattr(self, "_act_in_flight", None)
        if pending is not None and getattr(self, "_handle", None) is not None:  # collect before the handle goes away
            try:
                pending.item()
            except Exception:
                self._act_in_flight = None
        dp = self.__dict__.pop("_dp", None)  # (slimdqn/networks/parallel.py: the RCCL side of this handle goes first)
        if dp is not None:
            _hip.lib().idqn_dp_destroy(dp[0])
        if getattr(self, "_handle", None) is not None:
            _hip.lib().idqn_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:
            pass

    @staticmethod
    def _dev(x, dtype):
        t = getattr(x, "tensor", x)
        if not isinstance(t, torch.Tensor):
            a = np.ascontiguousarray(np.asarray(t))
            t = torch.from_numpy(a if a.flags.writeable else a.copy())  # (a read-only view, e.g. an environment's state table)
        return t.to(device="cuda", dtype=dtype).contiguous()

    # ---- the step ------------------------------------------------------------------------------------
    def _prepare(self, batch):
        """``(B, pointers)`` of a ReplayElement-like batch on the device: state, next_state, action, reward, terminal."""
        # The replay buffer hands out the SAME ReplayElement object (views of a staging set) every other sample: its
        # converted tensors and argument pointers are kept per object (a handful of entries; the entry holds the object).
        cache = self.__dict__.setdefault("_learn_cache", {})
        hit = cache.get(id(batch))
        if hit is not None and hit[0] is batch and self._handle is not None and hit[1] <= self._handle_batch:
            return hit[1], hit[2]
        if self._arch == "cnn":
            s, s2 = self._dev(batch.state, torch.uint8), self._dev(batch.next_state, torch.uint8)
            assert tuple(s.shape[1:]) == self._obs, f"state shape {tuple(s.shape)} vs observation_dim {self._obs}"
        else:
            s, s2 = self._dev(batch.state, torch.float32), self._dev(batch.next_state, torch.float32)
            assert int(np.prod(s.shape[1:])) == self._obs[0], f"state shape {tuple(s.shape)} vs dim {self._obs[0]}"
        B = int(s.shape[0])
        a = self._dev(batch.action, torch.int32)
        r = self._dev(batch.reward, torch.float32)
        t = self._dev(batch.is_terminal, torch.uint8)
        assert a.numel() == B and r.numel() == B and t.numel() == B
        self._ensure_handle(B)
        self._keep = (s, s2, a, r, t)  # keep inputs alive until the stream has consumed them
        ptrs = (_hip.ptr(s), _hip.ptr(s2), _hip.ptr(a), _hip.ptr(r), _hip.ptr(t))
        if type(batch).__name__ == "ReplayElement" and all(hasattr(f, "tensor") for f in (batch.state, batch.action)):
            if len(cache) >= 8:
                cache.clear()
            if all(x.data_ptr() == getattr(f, "tensor", f).data_ptr() for x, f in
                   ((s, batch.state), (s2, batch.next_state), (a, batch.action), (r, batch.reward), (t, batch.is_terminal))):
                cache[id(batch)] = (batch, B, ptrs)
        return B, ptrs

    def _learn(self, batch, flags=0, mean_divisor=None):
        """One gradient step on a ReplayElement-like batch; returns the per-head losses (device, [K])."""
        B, ptrs = self._prepare(batch)
        _hip.check(_hip.lib().idqn_learn_on_batch(self._handle, *ptrs, B, int(mean_divisor or B), int(flags),
                                                  _hip.current_stream()), "idqn_learn_on_batch")
        return self._losses

    # ``update_online_params`` = ``replay_buffer.sample()`` + ``learn_on_batch`` (idqn.py:65-72, dqn.py:41-47).  On this package's
    # ReplayBuffer with Atari-shaped uint8 frames the two halves are ONE C call (``idqn_learn_on_replay``): the sampler draws the
    # same keys from the same generator, the stacked gather happens inside the step's staging launch and the minibatch is never
    # materialised.  Anything else (other buffers, shapes, the f32 conv mode) samples, gathers and learns as two calls.
    fuse_replay_sampling = True

    def _sample_and_learn(self, replay_buffer):
        rb = replay_buffer
        if not (self.fuse_replay_sampling and type(self)._learn is DeviceAgent._learn and self._arch == "cnn" and hasattr(rb, "sample_slots") and hasattr(rb, "ring_view")
                and getattr(self, "_replay_fused_ok", True) and rb._batch_size <= 256 and os.environ.get("IDQN_LEARN_ON_REPLAY", "1") != "0"):
            return self._learn(rb.sample())
        slots = rb.sample_slots()
        frames, n_frames, frame_bytes, rows, stack, fshape, fdt = rb.ring_view()
        if not (stack == 4 and self._obs[2] == 4 and tuple(fshape) == tuple(self._obs[:2]) and np.dtype(fdt) == np.uint8 and frame_bytes % 16 == 0):
            self._replay_fused_ok = False
            return self._learn(rb._gather(slots))
        B = int(slots.size)
        self._ensure_handle(B)
        slots = np.ascontiguousarray(slots, np.int32)
        rc = _hip.lib().idqn_learn_on_replay(self._handle, _hip.ptr(frames), int(n_frames), int(frame_bytes), _hip.ptr(rows),
                                             slots.ctypes.data, B, int(stack), B, 0, _hip.current_stream())
        if rc == _hip.E_INVALID and self.__dict__.get("_replay_fused_ok") is None:
            # this handle runs another conv path (IDQN_CONV=f32 / the general shapes): same slots, two calls, from now on
            self._replay_fused_ok = False
            return self._learn(rb._gather(slots))
        _hip.check(rc, "idqn_learn_on_replay")
        self._replay_fused_ok = True
        return self._losses

    def _local_target_update(self):
        """target <- online (real copy), then online[k] <- online[k+1] over THIS agent's heads (idqn.py:78-80)."""
        self._ensure_handle(32)
        _hip.check(_hip.lib().idqn_target_update(self._handle, _hip.current_stream()), "idqn_target_update")

    def _local_target_sync(self):
        """target[k] <- online[k-1], k >= 1, over THIS agent's heads (idqn.py:20-24)."""
        self._ensure_handle(32)
        _hip.check(_hip.lib().idqn_target_sync(self._handle, _hip.current_stream()), "idqn_target_sync")

    def _apply_adam(self):
        _hip.check(_hip.lib().idqn_apply_adam(self._handle, _hip.current_stream()), "idqn_apply_adam")

    def _q_values(self, which, head, state):
        """Q-values [n, A] (device) of one head for <= 32 states."""
        dt = torch.uint8 if self._arch == "cnn" else torch.float32
        s = self._dev(state, dt)
        n = 1 if s.numel() == int(np.prod(self._obs)) else int(s.shape[0])
        self._ensure_handle(32)
        self._keep_q = s
        _hip.check(_hip.lib().idqn_q_values(self._handle, int(which), int(head), _hip.ptr(s), n, _hip.ptr(self._q_out),
                                            _hip.current_stream()), "idqn_q_values")
        return self._q_out[:n]

    def _best_action(self, which, head, state):
        """Greedy action (device int32 scalar, first maximum on ties) of one head for one state: one C call.
        A host state goes through a pinned staging buffer (a pageable 28 KB upload costs ~100 us, this ~10)."""
        dt = torch.uint8 if self._arch == "cnn" else torch.float32
        st = getattr(state, "tensor", state)
        if isinstance(st, torch.Tensor) and st.is_cuda:
            s = self._dev(st, dt)
        else:
            # a host state: ONE C call uploads it from pinned memory, runs the single-state path, brings the action back
            # and synchronises (what the reference's select_action does with its blocking `.item()`)
            if not hasattr(self, "_act_pin"):
                n = int(np.prod(self._obs))
                self._act_pin = torch.empty(n, dtype=dt).pin_memory()
                self._act_pin_np = self._act_pin.numpy()
                self._act_out = torch.zeros(4, dtype=torch.int32).pin_memory()
                self._act_out_np = self._act_out.numpy()
            src = np.asarray(getattr(state, "tensor", state))
            assert src.size == self._act_pin_np.size, "best_action takes a single state"
            pending = getattr(self, "_act_in_flight", None)
            if pending is not None:  # (a lazy action nobody collected: finish it before the staging buffer is rewritten)
                pending.item()
            self._act_pin_np[:] = src.reshape(-1)  # casts like the array conversion of the reference's jit would
            self._ensure_handle(32)
            if self.lazy_host_actions:
                _hip.check(_hip.lib().idqn_act_host_begin(self._handle, int(which), int(head), C.c_void_p(self._act_pin.data_ptr()),
                                                          _hip.ptr(self._q_out), C.c_void_p(self._act_out.data_ptr()),
                                                          _hip.current_stream()), "idqn_act_host_begin")
                self._act_in_flight = _PendingHostAction(self)
                return self._act_in_flight
            _hip.check(_hip.lib().idqn_act_host(self._handle, int(which), int(head), C.c_void_p(self._act_pin.data_ptr()),
                                                _hip.ptr(self._q_out), C.c_void_p(self._act_out.data_ptr()),
                                                _hip.current_stream()), "idqn_act_host")
            return _HostAction(int(self._act_out_np[0]))
        assert s.numel() == int(np.prod(self._obs)), "best_action takes a single state"
        self._ensure_handle(32)
        self._keep_q = s
        if not hasattr(self, "_action_out"):
            self._action_out = torch.zeros(32, dtype=torch.int32, device="cuda")
        _hip.check(_hip.lib().idqn_best_action(self._handle, int(which), int(head), _hip.ptr(s), 1, _hip.ptr(self._q_out),
                                               _hip.ptr(self._action_out), _hip.current_stream()), "idqn_best_action")
        return self._action_out[0]

    def _debug(self, name):
        """Internal activation buffer as a flat float32 device tensor (tests only)."""
        p, nbytes = C.c_void_p(), C.c_int64()
        _hip.check(_hip.lib().idqn_debug_buffer(self._handle, name.encode(), C.byref(p), C.byref(nbytes)), name)
        out = torch.empty(nbytes.value // 4, dtype=torch.float32, device="cuda")
        C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(out.data_ptr()), p, C.c_size_t(nbytes.value), 3)
        return out

    def _numpy_tree(self, arena):
        host = arena.cpu().numpy()
        tree = {}
        for name, off, shape in self._leaves:
            mod, leaf = name.split("/")
            v = host[:, off : off + int(np.prod(shape))].reshape((self._K,) + tuple(shape)).copy()
            tree.setdefault(mod, {})[leaf] = v if self._stacked else v[0]
        return {"params": tree}

    def _load_flat(self, arena, flat):
        """flat: {"Conv_0/kernel": array [K, ...]} -> arena (fixture injection in tests)."""
        host = arena.cpu().numpy()
        for name, off, shape in self._leaves:
            v = np.asarray(flat[name], np.float32)
            if not self._stacked and v.shape == tuple(shape):
                v = v[None]
            host[:, off : off + int(np.prod(shape))] = v.reshape(self._K, -1)
        arena.copy_(torch.from_numpy(host))

    def _flat(self, arena):
        host = arena.cpu().numpy()
        return {name: host[:, off : off + int(np.prod(shape))].reshape((self._K,) + tuple(shape)).copy()
                for name, off, shape in self._leaves}

    def _flat_grad(self):
        """{"Conv_0/kernel": array [K, ...]} read from the two-region gradient arena (tests)."""
        host = self._grad.cpu().numpy()
        K, gP, (w0b, w0e) = self._K, self._gP, self._w0
        out = {}
        for name, off, shape in self._leaves:
            n = int(np.prod(shape))
            if w0e > w0b and off == w0b:
                base = K * gP + 64
                v = np.stack([host[base + k * (w0e - w0b) : base + k * (w0e - w0b) + n] for k in range(K)])
            else:
                o = off if off < w0b or w0e == w0b else off - (w0e - w0b)
                v = np.stack([host[k * gP + o : k * gP + o + n] for k in range(K)])
            out[name] = v.reshape((K,) + tuple(shape)).copy()
        return out

    def get_model(self):
        """Picklable ``{"params": self.params}`` with numpy leaves (idqn.py:133-134, experiments/base/utils.py:134).
        ``self.params`` is already the flax variables dict ``{"params": {"Conv_0": ...}}``, so the pickle holds
        ``model["params"]["params"]["Dense_0"]["kernel"]`` -- exactly what the reference's checkpoints hold."""
        return {"params": self._numpy_tree(self._online)}
