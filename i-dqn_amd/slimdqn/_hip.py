"""ctypes binding of libidqn_hip.so (include/idqn_hip.h) -- the only door to the device code.

There is NO host fallback: if the shared library is missing or a call fails, this module raises.
PyTorch-ROCm is used for device storage only (``tensor.data_ptr()``), never for arithmetic on the
hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IDQN_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libidqn_hip.so")  # override: kernel experiments

IDQN_ARCH_CNN, IDQN_ARCH_FC = 0, 1
IDQN_MAX_FEATURES, IDQN_MAX_LEAVES = 8, 24
F_GRADS_ONLY, F_PROFILE, F_STOP_AFTER_DENSE0, F_STOP_BEFORE_DENSE0_WGRAD, F_PROFILE_ALL = 1, 2, 4, 8, 16
FACTORED_DENSE0, FACTORED_REST = 1, 2
DP_SIDE_STREAM, DP_UNIQUE_ID_BYTES = 1, 128
E_INVALID, E_HIP, E_RANGE, E_ASSERT = -1, -2, -3, -4


class HipExtensionError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("arch", C.c_int32), ("n_heads", C.c_int32), ("n_actions", C.c_int32),
        ("obs_h", C.c_int32), ("obs_w", C.c_int32), ("obs_c", C.c_int32),
        ("n_features", C.c_int32), ("features", C.c_int32 * IDQN_MAX_FEATURES),
        ("max_batch", C.c_int32),
        ("learning_rate", C.c_double), ("adam_b1", C.c_double), ("adam_b2", C.c_double), ("adam_eps", C.c_double),
        ("gamma_n", C.c_double),
        ("n_quantiles", C.c_int32), ("reserved", C.c_int32),
    ]


class Leaf(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("offset", C.c_int64), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


# every exported symbol of include/idqn_hip.h: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "idqn_last_error": (C.c_char_p, []),
    "idqn_abi_version": (C.c_int, []),
    "idqn_layout": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_int32), C.POINTER(Leaf), C.POINTER(C.c_int64)]),
    "idqn_create": (C.c_int, [C.POINTER(Config), _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_P)]),
    "idqn_destroy": (C.c_int, [_P]),
    "idqn_learn_on_batch": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_uint32, _P]),
    "idqn_learn_on_replay": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, _P]),
    "idqn_iqn_learn_on_batch": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_uint32, _P]),
    "idqn_iqn_q_values": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P, _P]),
    "idqn_backward_rest": (C.c_int, [_P, _P]),
    "idqn_export_dense0_factors": (C.c_int, [_P, _P, _P, _P]),
    "idqn_dense0_factors": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "idqn_finish_step_factored": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32] + [C.c_int64] * 6 + [C.c_uint32, _P]),
    "idqn_apply_adam": (C.c_int, [_P, _P]),
    "idqn_dp_unique_id": (C.c_int, [_P]),
    "idqn_dp_create": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(_P)]),
    "idqn_dp_create_from_comm": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(_P)]),
    "idqn_dp_destroy": (C.c_int, [_P]),
    "idqn_dp_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "idqn_dp_step": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_uint32, _P]),
    "idqn_set_per_buffers": (C.c_int, [_P, _P, _P]),
    "sumtree_set_one": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_double, _P, _P]),
    "sampler_mailbox_create": (C.c_int, [C.c_int32, C.POINTER(_P)]),
    "sampler_mailbox_destroy": (C.c_int, [_P]),
    "sumtree_query_host": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P, C.POINTER(C.c_double),
                                    C.POINTER(C.c_int32), _P]),
    "sampler_map_set": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "sampler_map_indices": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "sampler_prioritized_add": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_double, _P]),
    "sampler_prioritized_remove": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, _P]),
    "per_sample_leaves": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]),
    "per_importance_weights": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int64, C.c_double, _P, _P]),
    "per_priorities_from_td": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, _P, _P, _P]),
    "idqn_target_update": (C.c_int, [_P, _P]),
    "idqn_target_sync": (C.c_int, [_P, _P]),
    "idqn_q_values": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P]),
    "idqn_best_action": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P]),
    "idqn_act_host": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "idqn_act_host_begin": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "idqn_act_host_end": (C.c_int, [_P, _P, _P]),
    "idqn_debug_buffer": (C.c_int, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "idqn_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_char_p]),
    "idqn_profile_table": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "sumtree_set": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P]),
    "sumtree_get": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P]),
    "sumtree_query": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, _P]),
    "replay_gather_stacked": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P, _P, _P, _P,
                                       _P, _P]),
    "replay_gather": (C.c_int, [_P, C.c_int64, _P, C.c_int32, _P, _P, _P]),
    "replay_add_frame": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "replay_ring_regrow": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P]),
    "replay_gather_scalars": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P, _P, _P, _P]),
}

_lib = None


def lib():
    """Loads the extension (once).  Raises HipExtensionError if it was not built."""
    global _lib
    if _lib is None:
        # torch first: its wheel bundles the HIP runtime (libamdhip64) the process must share -- loading
        # libidqn_hip.so before torch would bring in a second runtime and the later one finds no device
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what=""):
    """Maps C return codes to the exception types the reference raises."""
    if rc == 0:
        return
    msg = lib().idqn_last_error().decode(errors="replace")
    if rc == E_RANGE:
        raise ValueError(msg or what)
    if rc == E_ASSERT:
        raise AssertionError(msg or what)
    raise HipExtensionError(f"{what} failed with code {rc}: {msg}")


def ptr(t):
    """Device address of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


class _RawDeviceArray:
    """Device memory owned by the library, presented through the CUDA array interface so that torch can alias it."""

    def __init__(self, address, n_floats):
        self.__cuda_array_interface__ = {"shape": (int(n_floats),), "typestr": "<f4", "data": (int(address), False),
                                         "version": 2, "strides": None}


def device_view(address, n_floats):
    """float32 torch tensor over `n_floats` floats of library-owned device memory at `address` (no copy)."""
    import torch

    return torch.as_tensor(_RawDeviceArray(address, n_floats), device="cuda")


def current_stream():
    """torch's current stream of the current device as a hipStream_t.  (Goes to the raw-stream call directly:
    ``torch.cuda.current_stream()`` resolves the device index through several Python layers, 12 us per call, and the
    step makes several calls.)"""
    import torch

    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def make_config(arch, n_heads, n_actions, obs, features, max_batch, lr, eps, gamma_n, b1=0.9, b2=0.999, n_quantiles=0):
    cfg = Config()
    cfg.arch = {"cnn": IDQN_ARCH_CNN, "fc": IDQN_ARCH_FC}[arch]
    cfg.n_heads, cfg.n_actions = int(n_heads), int(n_actions)
    cfg.obs_h, cfg.obs_w, cfg.obs_c = (int(x) for x in obs)
    cfg.n_features = len(features)
    for i, f in enumerate(features):
        cfg.features[i] = int(f)
    cfg.max_batch = int(max_batch)
    cfg.learning_rate, cfg.adam_b1, cfg.adam_b2, cfg.adam_eps, cfg.gamma_n = float(lr), b1, b2, float(eps), float(gamma_n)
    cfg.n_quantiles = int(n_quantiles)
    return cfg


def layout(cfg):
    """[(name, offset, shape)], head_stride -- needs no GPU."""
    n = C.c_int32()
    leaves = (Leaf * IDQN_MAX_LEAVES)()
    stride = C.c_int64()
    check(lib().idqn_layout(C.byref(cfg), C.byref(n), leaves, C.byref(stride)), "idqn_layout")
    out = [(leaves[i].name.decode(), int(leaves[i].offset), tuple(int(s) for s in leaves[i].shape[: leaves[i].ndim]))
           for i in range(n.value)]
    return out, int(stride.value)
