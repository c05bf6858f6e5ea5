"""GPU: the data-parallel step issued by ONE C call (include/idqn_hip.h: idqn_dp_*, csrc/dp.hip) -- the library's own RCCL
communicator, or one the caller owns -- against the Python schedule over torch.distributed (slimdqn/networks/parallel.py), which
is its oracle.  One rank (RCCL refuses two ranks per GPU): what is under test is the call sequence, the stream ordering (the
stress in tests/test_gpu_configs.py runs 200 steps of it in both stream modes) and the entry points themselves."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    import torch
    import torch.distributed as dist

    owns = not dist.is_initialized()
    if owns:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    if owns:
        dist.destroy_process_group()


def _agents_and_batches(K=2, B=64, n=2):
    import torch
    from collections import namedtuple

    from oracle import qnet_ref as Q
    from slimdqn.networks.idqn import iDQN

    arch, obs, A, feats = "cnn", (84, 84, 4), 6, [32, 64, 64, 512]
    Batch = namedtuple("Batch", "state action reward next_state is_terminal")
    batches = [Batch(*(torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in Q.synthetic_batch(70 + i, B, obs, A, arch))) for i in range(3)]
    agents = [iDQN(0, obs, A, K, feats, arch, 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4) for _ in range(n)]
    return agents, batches


def test_step_on_a_communicator_the_caller_owns(rccl_group):
    """idqn_dp_create_from_comm: an ncclComm_t made outside the library (here straight from librccl through ctypes) carries the
    step; idqn_dp_info reports its rank / world and the bytes of the two collectives; results bit-identical to the Python
    schedule; idqn_dp_destroy leaves the caller's communicator alive."""
    import torch

    from slimdqn import _hip
    from slimdqn.networks.parallel import data_parallel_step

    lib = _hip.lib()
    (a, b), batches = _agents_and_batches()

    class Uid(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    rccl = C.CDLL("librccl.so.1")
    uid = Uid()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    Bsz, ptrs = a._prepare(batches[0])
    dp = C.c_void_p()
    _hip.check(lib.idqn_dp_create_from_comm(a._handle, comm, 0, C.byref(dp)), "idqn_dp_create_from_comm")
    rank, world, gb, ab = C.c_int32(-1), C.c_int32(-1), C.c_int64(), C.c_int64()
    _hip.check(lib.idqn_dp_info(dp, C.byref(rank), C.byref(world), C.byref(gb), C.byref(ab)), "idqn_dp_info")
    assert (rank.value, world.value) == (0, 1)
    assert gb.value == 2 * (7744 + 512) * 32 * 4  # K x (F + J) x 32 floats per 32-sample block
    assert ab.value == a._grad_small.numel() * 4
    for i in range(6):
        Bsz, ptrs = a._prepare(batches[i % 3])
        _hip.check(lib.idqn_dp_step(dp, *ptrs, Bsz, Bsz, 0, _hip.current_stream()), "idqn_dp_step")
        data_parallel_step(b, batches[i % 3], Bsz, mode="factored", serial=True)
    torch.cuda.synchronize()
    for x, y in ((a._online, b._online), (a._mu, b._mu), (a._nu, b._nu), (a._losses, b._losses)):
        assert (x.cpu().numpy().view(np.uint32) == y.cpu().numpy().view(np.uint32)).all()
    assert (a._count == b._count).all() and (a._cum.cpu().numpy() == b._cum.cpu().numpy()).all()
    # refused: a global batch that is not world x shard, flags other than the profile ones
    assert lib.idqn_dp_step(dp, *ptrs, Bsz, 2 * Bsz, 0, _hip.current_stream()) == _hip.E_INVALID
    assert lib.idqn_dp_step(dp, *ptrs, Bsz, Bsz, _hip.F_GRADS_ONLY, _hip.current_stream()) == _hip.E_INVALID
    _hip.check(lib.idqn_dp_destroy(dp), "idqn_dp_destroy")
    n = C.c_int(-1)
    assert rccl.ncclCommCount(comm, C.byref(n)) == 0 and n.value == 1  # still the caller's
    assert rccl.ncclCommDestroy(comm) == 0
    for ag in (a, b):
        ag._destroy_handle()


def test_create_checks_its_arguments(rccl_group):
    from slimdqn import _hip
    from slimdqn.networks.idqn import iDQN

    lib = _hip.lib()
    uid = (C.c_ubyte * _hip.DP_UNIQUE_ID_BYTES)()
    _hip.check(lib.idqn_dp_unique_id(uid), "idqn_dp_unique_id")
    assert any(uid)  # RCCL filled it in
    fc = iDQN(0, 8, 4, 3, [100, 100], "fc", 3e-4, 0.99, 1, 1, 200, 10)
    fc._ensure_handle(32)
    dp = C.c_void_p()
    assert lib.idqn_dp_create(fc._handle, uid, 0, 1, 0, C.byref(dp)) == _hip.E_INVALID  # the step is built for the MFMA cnn path
    assert b"cnn" in lib.idqn_last_error()
    (a,), _ = _agents_and_batches(n=1)
    a._ensure_handle(32)
    assert lib.idqn_dp_create(a._handle, uid, 1, 1, 0, C.byref(dp)) == _hip.E_INVALID  # rank outside the world
    assert lib.idqn_dp_create(a._handle, uid, 0, 1, 8, C.byref(dp)) == _hip.E_INVALID  # unknown flag
    assert lib.idqn_dp_destroy(None) == 0
    a._destroy_handle()
