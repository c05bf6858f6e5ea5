"""Launch-structure switches of the HIP path must not change results: the same seeded steps are run in child processes
(the switches are read once per process) and compared with the default build of the step.

  IDQN_STEP_GRAPH=1   the plain step replayed as a hipGraph          -> bit-identical to the eager launches
  IDQN_CONV_CHAIN=1   the three forward convs as ONE launch with per-item flag hand-offs (csrc/convp_chain.hip)
                      -> bit-identical over 60 steps (flags only, no sum changes its order), no spin gave up (finite losses)
  IDQN_D0_GROUP=1 / IDQN_D0_FUSE_HIDDEN=1   the Dense_0 forward with the in-workgroup split reduction / also with the head's
                      first stage riding in it (last-arriver hand-off)  -> fp32 round-off / bit-identical to the grouped one
  IDQN_D0_PAIR=0 / 2  the fused Dense_0 update on single tiles + k_da3_finalize / on pairs of row tiles + finalize instead of
                      pairs of column tiles with dL/da3 finished in the kernel (default)  -> bit-identical
  IDQN_D0_FWD_DMA=1   the Dense_0 forward fed by per-wave LDS-DMA rings instead of vector registers  -> bit-identical
  IDQN_DP_ALDS=0/1/2  the factored data-parallel update's contraction: registers / a3 fragments through LDS / the same on
                      64 x 256 tiles  -> bit-identical
  IDQN_D0_FIN=1       the last-arriving column-tile workgroup instead of the k_da3_finalize launch  -> bit-identical
  IDQN_ADAM_ROLE=1    the Conv_0 weight-gradient launch carries the other small leaves' Adam update  -> bit-identical at
                      equal chunk counts
  IDQN_NO_PAIR=1      conv data / weight gradients as two launches   -> same losses, parameters within fp32 round-off
                      (the weight gradient is cut into a different number of position chunks, i.e. summed in another order)
  IDQN_ACT_POLL=0     acting result by copy + synchronisation        -> same greedy actions as the polled mailbox
  IDQN_OVERLAP=1      the last items of the fused Dense_0 update run as stream roles of the Conv_2 pair and Conv_0
                      weight-gradient launches (csrc/dense0_update.h)   -> same losses, Dense_0 bit-identical, conv leaves
                      within fp32 round-off (those launches are planned for fewer workgroups = other chunk sums)
  IDQN_D0_ROWS=1      the fused Dense_0 kernel on whole rows, no finalize launch  -> same, conv leaves within round-off
  IDQN_IQN_GEMM=0     the i-IQN heads' Dense_0 on the plain step's per-block kernels instead of the tiled GEMMs
                      (csrc/iqn_gemm.h)  -> first-step losses bit-identical (the forward sums in the same order), parameters
                      within fp32 round-off after a few steps
  IDQN_FC_NO_MFMA_G=1 / IDQN_FC_NO_MFMA=1 / IDQN_FC_GENERIC=1   the MLP step on the LDS kernel / the generic kernel instead of
                      the MFMA kernels (csrc/fc_kernels.h)  -> same losses and parameters within fp32 round-off
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, os
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from collections import namedtuple
from slimdqn.networks.idqn import iDQN
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
rng = np.random.default_rng(7)
agent = iDQN(3, (84, 84, 4), 6, 5, [32, 64, 64, 512], "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
B = int(os.environ.get("SW_BATCH", "32"))
def batch():
    return Batch(torch.from_numpy(rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)).cuda(),
                 torch.from_numpy(rng.integers(0, 6, B).astype(np.int32)).cuda(),
                 torch.from_numpy(rng.standard_normal(B).astype(np.float32)).cuda(),
                 torch.from_numpy(rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)).cuda(),
                 torch.from_numpy((rng.random(B) < 0.1).astype(np.uint8)).cuda())
bs = [batch(), batch()]  # two buffer sets, used in turn like the replay buffer's staging sets
losses = [agent._learn(bs[i % 2]).cpu().numpy().astype(np.float64).tolist() for i in range(int(os.environ.get("SW_STEPS", "6")))]
state = rng.integers(0, 256, (84, 84, 4), dtype=np.uint8)
acts = [int(agent._best_action(0, k, state)) for k in range(5)]
flat = agent._flat(agent._online)
probe = {name: v.reshape(5, -1)[:, :: max(1, v[0].size // 97)].astype(np.float64).tolist() for name, v in flat.items()}
print("RESULT" + json.dumps({"losses": losses, "acts": acts, "probe": probe}))
"""


# Switches the shipped library reads itself; every other IDQN_* switch exists only in the -DIDQN_VARIANTS build
# (i-dqn_amd/libidqn_hip_variants.so, built by __graft_entry__.build()), which the child then loads through IDQN_HIP_LIB.
SHIPPED = {"IDQN_STEP_GRAPH", "IDQN_ACT_POLL", "IDQN_ACT_GRAPH", "IDQN_ACT_GENERIC", "IDQN_CONV", "IDQN_CNN_GENERAL",
           "IDQN_PLAN_PRINT", "IDQN_LOOP_OVERLAP", "IDQN_DP_MODE", "IDQN_DP_OVERLAP"}
VARIANTS_LIB = os.path.join(ROOT, "i-dqn_amd", "libidqn_hip_variants.so")


def _child_env(env):
    e = dict(os.environ)
    e.update(env)
    if any(k.startswith("IDQN_") and k not in SHIPPED for k in env):
        assert os.path.exists(VARIANTS_LIB), "run __graft_entry__.build() first: it also builds the variants library"
        e["IDQN_HIP_LIB"] = VARIANTS_LIB
    return e


def _run(**env):
    e = _child_env(env)
    out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):])


@pytest.fixture(scope="module")
def default_run():
    return _run()


def test_step_graph_replay_is_bit_identical(default_run):
    got = _run(IDQN_STEP_GRAPH="1")
    assert got["losses"] == default_run["losses"]
    assert got["probe"] == default_run["probe"]


def test_chained_forward_convs_are_bit_identical():
    # 60 steps: every hand-off of every step has to deliver the producer's bytes (a stale or early read changes the bits
    # of everything downstream); the losses would be NaN had a bounded spin given up (k_td_dh reads the chain's err word)
    want = _run(SW_STEPS="60")
    got = _run(SW_STEPS="60", IDQN_CONV_CHAIN="1")
    assert np.isfinite(np.asarray(got["losses"])).all()
    assert got["losses"] == want["losses"]
    assert got["probe"] == want["probe"]
    assert got["acts"] == want["acts"]


def test_dense0_forward_group_and_fused_head_stage(default_run):
    """Round 4 (both opt-in, measured neutral): IDQN_D0_GROUP=1 lets the Dense_0 forward add four consecutive splits per
    workgroup through LDS (a quarter of the partial slabs; another association of the same sum -> fp32 round-off);
    IDQN_D0_FUSE_HIDDEN=1 additionally lets that launch carry the head's first stage, the last-arriving workgroup of a
    column tile doing what k_hidden does -> bit-identical to the grouped forward + k_hidden."""
    got = _run(IDQN_D0_GROUP="1")
    fused = _run(IDQN_D0_FUSE_HIDDEN="1")
    assert fused["losses"] == got["losses"]
    assert fused["probe"] == got["probe"]
    np.testing.assert_allclose(np.asarray(got["losses"]), np.asarray(default_run["losses"]), rtol=0, atol=1e-6)
    for name, want in default_run["probe"].items():
        np.testing.assert_allclose(np.asarray(got["probe"][name]), np.asarray(want), rtol=0, atol=2e-6, err_msg=name)


def test_dense0_update_finishes_the_data_gradient_itself(default_run):
    """Round 4 (opt-in, measured slower): IDQN_D0_FIN=1 lets the column-tile workgroup whose partial data gradient arrives
    last add the tiles in tile order, apply the ReLU mask and write the planes / per-position sums (DenseWgradArgs::fin_ctr)
    instead of the k_da3_finalize launch.  Same sums in the same order: bit-identical."""
    got = _run(IDQN_D0_FIN="1")
    assert got["losses"] == default_run["losses"]
    assert got["probe"] == default_run["probe"]
    assert got["acts"] == default_run["acts"]


def test_dense0_update_on_pairs_of_column_tiles_is_bit_identical(default_run):
    """Round 4 (default): one workgroup takes both column tiles of its 32 rows, one after the other; it adds the two partial data
    gradients itself (tile 0 + tile 1, k_da3_finalize's order), masks and writes the output forms -- no partials in HBM, no
    finalize launch.  IDQN_D0_PAIR=0 is the tile kernel + k_da3_finalize, =2 pairs of row tiles + finalize.  Same arithmetic in
    the same order everywhere: bit-identical."""
    for mode in ("0", "2", "3"):  # (3: the pair kernel with whole tiles in flight and cross-tile refills)
        got = _run(IDQN_D0_PAIR=mode)
        assert got["losses"] == default_run["losses"], mode
        assert got["probe"] == default_run["probe"], mode
        assert got["acts"] == default_run["acts"], mode


def test_eight_block_data_gradient_finished_in_the_gemm_epilogue_is_bit_identical():
    """Round 5 (groups of 8 sample blocks, B = 256): the Dense_0 data gradient is the tiled GEMM writing raw rows + k_da3_finalize.
    IDQN_NB_DGRAD_FIN=1 (opt-in, measured neutral) masks, splits into planes and sums per position in the GEMM's epilogue: same
    values, same summation order -> bit-identical.  (IDQN_NB_DGRAD_F32=1, the f32-MFMA kernel per block, is another
    association: fp32 round-off.)"""
    want = _run(SW_BATCH="256", SW_STEPS="3")
    got = _run(SW_BATCH="256", SW_STEPS="3", IDQN_NB_DGRAD_FIN="1")
    assert got["losses"] == want["losses"]
    assert got["probe"] == want["probe"]
    f32 = _run(SW_BATCH="256", SW_STEPS="3", IDQN_NB_DGRAD_F32="1")
    np.testing.assert_allclose(np.asarray(f32["losses"]), np.asarray(want["losses"]), rtol=0, atol=1e-6)
    for name, w in want["probe"].items():
        np.testing.assert_allclose(np.asarray(f32["probe"][name]), np.asarray(w), rtol=0, atol=2e-6, err_msg=name)


def test_dense0_cache_policy_changes_no_bit(default_run):
    """Round 5 (default while the online nets' Dense_0 kernels fit the memory-side cache, K <= 5): the forward reads them with
    default-policy loads and the fused update stores theta_new with the default policy (csrc/qnet.hip d0_keep_online).
    IDQN_D0_KEEP=0 is rounds 3-4's policy (every Dense_0 stream non-temporal), IDQN_D0_FWD_NT_FROM=10 every net default-policy:
    a cache policy changes no arithmetic -> bit-identical."""
    # (IDQN_D0_KEEP=3: the update as two launches, heads 0-2 with the default-policy store, heads 3-4 non-temporal; IDQN_D0_KEEP_ALL=1:
    # every stream of the update default-policy, the K <= 2 default)
    for env in ({"IDQN_D0_KEEP": "0"}, {"IDQN_D0_KEEP": "3"}, {"IDQN_D0_KEEP_ALL": "1"}, {"IDQN_D0_FWD_NT_FROM": "10"}, {"IDQN_D0_NET_ROT": "0"}):
        got = _run(**env)
        assert got["losses"] == default_run["losses"], env
        assert got["probe"] == default_run["probe"], env
        assert got["acts"] == default_run["acts"], env


def test_dense0_forward_through_lds_dma_is_bit_identical(default_run):
    """Round 4 (opt-in, measured neutral): IDQN_D0_FWD_DMA=1 sends the Dense_0 forward's weight stream and activations through
    per-wave LDS-DMA rings (k_dense0_fwd3d) instead of vector registers.  Same k order, splits and product order: bit-identical."""
    got = _run(IDQN_D0_FWD_DMA="1")
    assert got["losses"] == default_run["losses"]
    assert got["probe"] == default_run["probe"]
    assert got["acts"] == default_run["acts"]


CHILD_DP = r"""
import json, sys, os
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from collections import namedtuple
from slimdqn import _hip
from slimdqn.networks.idqn import iDQN
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
rng = np.random.default_rng(11)
N = int(os.environ["SW_RANKS"])
agent = iDQN(3, (84, 84, 4), 6, 5, [32, 64, 64, 512], "cnn", 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
def batch():
    return Batch(torch.from_numpy(rng.integers(0, 256, (32, 84, 84, 4), dtype=np.uint8)).cuda(),
                 torch.from_numpy(rng.integers(0, 6, 32).astype(np.int32)).cuda(),
                 torch.from_numpy(rng.standard_normal(32).astype(np.float32)).cuda(),
                 torch.from_numpy(rng.integers(0, 256, (32, 84, 84, 4), dtype=np.uint8)).cuda(),
                 torch.from_numpy((rng.random(32) < 0.1).astype(np.uint8)).cuda())
bs = [batch(), batch()]
lib, q, K = _hip.lib(), _hip.current_stream, agent._K
F, J = next(shape for name, _, shape in agent._leaves if name == "Dense_0/kernel")
X, Y = F * 32, J * 32
n_a3, n_dh = K * X, K * Y
send = torch.zeros(n_a3 + n_dh, dtype=torch.float32, device="cuda")
gathered = torch.zeros(N * (n_a3 + n_dh), dtype=torch.float32, device="cuda")
losses = []
for i in range(4):  # rank 0's calls of an N-rank factored step (slimdqn/networks/parallel.py); every slot holds this rank's factors
    agent._learn(bs[i % 2], flags=_hip.F_STOP_BEFORE_DENSE0_WGRAD, mean_divisor=32 * N)
    _hip.check(lib.idqn_export_dense0_factors(agent._handle, _hip.ptr(send), _hip.ptr(send[n_a3:]), q()), "export")
    for r in range(N):
        gathered[r * (n_a3 + n_dh) : (r + 1) * (n_a3 + n_dh)].copy_(send)
    _hip.check(lib.idqn_backward_rest(agent._handle, q()), "rest")
    fa = (agent._handle, _hip.ptr(gathered), _hip.ptr(gathered[n_a3:]), N, 1, n_a3 + n_dh, X, X, n_a3 + n_dh, Y, Y)
    _hip.check(lib.idqn_finish_step_factored(*fa, _hip.FACTORED_DENSE0, q()), "finish")
    _hip.check(lib.idqn_finish_step_factored(*fa, _hip.FACTORED_REST, q()), "finish")
    losses.append(agent._losses.cpu().numpy().astype(np.float64).tolist())
flat = agent._flat(agent._online)
probe = {name: v.reshape(5, -1)[:, :: max(1, v[0].size // 997)].astype(np.float64).tolist() for name, v in flat.items()}
print("RESULT" + json.dumps({"losses": losses, "probe": probe}))
"""


@pytest.mark.parametrize("ranks", [3, 9])
def test_factored_update_contraction_variants_are_bit_identical(ranks):
    """Round 4: the factored data-parallel update contracts N sample blocks per head.  Default (IDQN_DP_ALDS=1): the tile's a3
    fragments are staged once by LDS-DMA (chunks of 8 blocks: N = 9 takes two), the dh fragments run in a register ring;
    IDQN_DP_ALDS=2 the same on 64 x 256 tiles (chunks of 4 blocks); IDQN_DP_ALDS=0 the register version.  Every accumulator
    takes the same products in the same order: bit-identical parameters after 4 emulated N-rank steps."""
    def run(**env):
        e = _child_env(dict(env, SW_RANKS=str(ranks)))
        out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD_DP], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1]
        return json.loads(line[len("RESULT"):])
    ref = run(IDQN_DP_ALDS="0")
    assert np.isfinite(np.asarray(ref["losses"])).all()
    for mode in ("1", "2"):
        got = run(IDQN_DP_ALDS=mode)
        assert got["losses"] == ref["losses"], mode
        assert got["probe"] == ref["probe"], mode


def test_adam_role_of_the_conv0_weight_gradient_launch(default_run):
    """Round 4 (opt-in, measured slower): IDQN_ADAM_ROLE=1 lets the Conv_0 weight-gradient launch carry the Adam update of
    every other small leaf on the CUs it leaves free (default: one Adam launch for all small leaves behind a whole-chip weight
    gradient).  The same update arithmetic per element; Conv_0's gradient is cut into another number of position chunks."""
    got = _run(IDQN_ADAM_ROLE="1")
    np.testing.assert_allclose(np.asarray(got["losses"]), np.asarray(default_run["losses"]), rtol=0, atol=1e-6)
    for name, want in default_run["probe"].items():
        if name.startswith("Conv_0"):
            np.testing.assert_allclose(np.asarray(got["probe"][name]), np.asarray(want), rtol=0, atol=2e-6, err_msg=name)
    # with the same chunk count on both sides everything is bit-identical: the role only moves WHERE the update runs
    a, b = _run(IDQN_ADAM_ROLE="1"), _run(IDQN_WCHUNKS="32")
    assert a["losses"] == b["losses"]
    assert a["probe"] == b["probe"]


def test_unpaired_conv_backward_matches(default_run):
    got = _run(IDQN_NO_PAIR="1")
    np.testing.assert_allclose(np.asarray(got["losses"]), np.asarray(default_run["losses"]), rtol=0, atol=1e-6)
    for name, want in default_run["probe"].items():
        # six Adam steps at lr 6.25e-5 on gradients that differ in their last bits
        np.testing.assert_allclose(np.asarray(got["probe"][name]), np.asarray(want), rtol=0, atol=2e-6, err_msg=name)


def test_overlapped_dense0_update_matches(default_run):
    got = _run(IDQN_OVERLAP="1")
    np.testing.assert_allclose(np.asarray(got["losses"]), np.asarray(default_run["losses"]), rtol=0, atol=1e-6)
    for name, want in default_run["probe"].items():
        np.testing.assert_allclose(np.asarray(got["probe"][name]), np.asarray(want), rtol=0, atol=2e-6, err_msg=name)


def test_full_row_dense0_kernel_matches(default_run):
    """IDQN_D0_ROWS=1: the fused Dense_0 kernel on whole 512-column rows finishes dL/da3 itself (no finalize launch); the
    data gradient is summed in another order, so conv leaves agree to fp32 round-off, Dense_0 itself bit for bit."""
    got = _run(IDQN_D0_ROWS="1")
    np.testing.assert_allclose(np.asarray(got["losses"]), np.asarray(default_run["losses"]), rtol=0, atol=1e-6)
    for name, want in default_run["probe"].items():
        np.testing.assert_allclose(np.asarray(got["probe"][name]), np.asarray(want), rtol=0, atol=2e-6, err_msg=name)


def test_acting_by_copy_and_sync_matches(default_run):
    got = _run(IDQN_ACT_POLL="0")
    assert got["acts"] == default_run["acts"]
    assert got["losses"] == default_run["losses"]


TRAINER_CHILD = r"""
import json, sys, os, tempfile
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np
from experiments.atari.idqn import run
argv = ["-en", "g", "-s", "3", "-ne", "1", "-ntspe", "90", "-nis", "40", "-rbc", "120", "-nn", "2", "-at", "cnn",
        "-tuf", "20", "-tsf", "5", "-f", "32", "64", "64", "128", "-horizon", "30", "-bs", "32"]
p, agent = run(argv, save_root=tempfile.mkdtemp())
flat = agent._flat(agent._online)
tflat = agent._flat(agent._target)
probe = {name: v.reshape(2, -1)[:, :: max(1, v[0].size // 61)].astype(np.float64).tolist() for name, v in flat.items()}
tprobe = {name: v.reshape(2, -1)[:, :: max(1, v[0].size // 61)].astype(np.float64).tolist() for name, v in tflat.items()}
print("RESULT" + json.dumps({"probe": probe, "target": tprobe, "count": int(agent._count[0].item())}))
"""


def test_trainer_loop_with_and_without_step_graph():
    """The trainer's launcher switches IDQN_STEP_GRAPH on (experiments/base/launch.py).  The whole loop -- acting graphs,
    replay staging sets, gradient steps, T-step copies and shifts, D-step syncs in between -- must end with bit-identical
    online and target parameters whether the step is replayed as a graph or issued launch by launch."""
    def run(flag):
        e = dict(os.environ, IDQN_STEP_GRAPH=flag)
        out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + TRAINER_CHILD], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1][len("RESULT"):])

    a, b = run("1"), run("0")
    assert a["count"] == b["count"] and a["count"] >= 40
    assert a["probe"] == b["probe"]
    assert a["target"] == b["target"]


def test_trainer_loop_with_and_without_overlapped_replay_add():
    """Round 4: the launcher lets the greedy action's launch return at once (idqn_act_host_begin / _end) and runs the replay
    bookkeeping of the previous transition under it (ReplayBuffer.add_deferred; IDQN_LOOP_OVERLAP=0 switches both off).  Same
    transitions in the same order, same samples, same steps: bit-identical parameters at the end."""
    def run(flag):
        e = dict(os.environ, IDQN_LOOP_OVERLAP=flag)
        out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + TRAINER_CHILD], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1][len("RESULT"):])

    a, b = run("1"), run("0")
    assert a["count"] == b["count"] and a["count"] >= 40
    assert a["probe"] == b["probe"]
    assert a["target"] == b["target"]


FC_CHILD = r"""
import json, sys, os
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from collections import namedtuple
from slimdqn.networks.idqn import iDQN
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
out = {}
for name, K, feats, B in (("lunar", 3, [100, 100], 32), ("wide", 5, [200, 200], 48)):
    rng = np.random.default_rng(11)
    agent = iDQN(5, 8, 4, K, feats, "fc", 3e-4, 0.99, 1, 1, 10**9, 10**9)
    b = Batch(torch.from_numpy(rng.standard_normal((B, 8)).astype(np.float32)).cuda(),
              torch.from_numpy(rng.integers(0, 4, B).astype(np.int32)).cuda(),
              torch.from_numpy(rng.standard_normal(B).astype(np.float32)).cuda(),
              torch.from_numpy(rng.standard_normal((B, 8)).astype(np.float32)).cuda(),
              torch.from_numpy((rng.random(B) < 0.1).astype(np.uint8)).cuda())
    losses = [agent._learn(b).cpu().numpy().astype(np.float64).tolist() for _ in range(5)]
    flat = agent._flat(agent._online)
    out[name] = {"losses": losses, "probe": {n: v.reshape(K, -1)[:, :: max(1, v[0].size // 53)].astype(np.float64).tolist() for n, v in flat.items()}}
print("RESULT" + json.dumps(out))
"""


def test_mlp_kernel_variants_agree():
    """The MLP step has four kernels (MFMA with staged weights, MFMA with weights from global memory, LDS FMA, generic);
    which one runs is a matter of what fits LDS.  Forced onto the slower ones, the same seeded steps must give the same
    losses and parameters to fp32 round-off (the sums run in another order)."""
    def run(**env):
        e = _child_env(env)
        out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + FC_CHILD], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1][len("RESULT"):])

    base = run()
    for env in ({"IDQN_FC_NO_MFMA_G": "1"}, {"IDQN_FC_NO_MFMA": "1"}, {"IDQN_FC_GENERIC": "1"}):
        got = run(**env)
        for name in base:
            np.testing.assert_allclose(np.asarray(got[name]["losses"]), np.asarray(base[name]["losses"]), rtol=0, atol=2e-6, err_msg=str(env))
            for leaf, want in base[name]["probe"].items():
                np.testing.assert_allclose(np.asarray(got[name]["probe"][leaf]), np.asarray(want), rtol=0, atol=3e-6, err_msg=f"{env} {name} {leaf}")


IQN_CHILD = r"""
import json, sys, os
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from collections import namedtuple
from slimdqn.networks.iiqn import iIQN
Batch = namedtuple("Batch", "state action reward next_state is_terminal")
rng = np.random.default_rng(21)
obs, A, K, N, B = (20, 20, 4), 4, 2, int(os.environ.get("TEST_IQN_N", "16")), 32
agent = iIQN(9, obs, A, K, [32, 32, 32, 256], "cnn", 2.5e-4, 0.99, 1, 1, 10**9, 10**9, adam_eps=1e-6, n_quantiles=N)
b = Batch(rng.integers(0, 256, size=(B,) + obs, dtype=np.uint8), rng.integers(0, A, size=B).astype(np.int32),
          rng.standard_normal(B).astype(np.float32), rng.integers(0, 256, size=(B,) + obs, dtype=np.uint8), rng.random(B) < 0.1)
taus = rng.random((K, 3, N, B)).astype(np.float32) * 0.98 + 0.01
losses = [agent._learn(b, taus=taus).cpu().numpy().astype(np.float64).tolist() for _ in range(4)]
flat = agent._flat(agent._online)
probe = {n: v.reshape(K, -1)[:, :: max(1, v[0].size // 71)].astype(np.float64).tolist() for n, v in flat.items()}
print("RESULT" + json.dumps({"losses": losses, "probe": probe}))
"""


def test_iqn_gemm_kernels_match_the_per_block_kernels():
    """N = 16 fraction blocks: all three Dense_0 GEMMs of the quantile heads run by default; IDQN_IQN_GEMM=0 sends the same
    step through the per-block kernels of the plain step.  The forward GEMM keeps that kernel's summation order (first
    loss bit-identical); the gradients sum in another order and the data gradient moves from the f32 to the split-bf16
    products: parameters agree to fp32 round-off."""
    def run(**env):
        e = _child_env(env)
        out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + IQN_CHILD], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1][len("RESULT"):])

    # N = 8: forward and data gradient as GEMMs, the weight gradient on the fused kernel of the plain step (it needs 16 blocks)
    a8, b8 = run(TEST_IQN_N="8"), run(TEST_IQN_N="8", IDQN_IQN_GEMM="0")
    assert a8["losses"][0] == b8["losses"][0]
    np.testing.assert_allclose(np.asarray(a8["losses"]), np.asarray(b8["losses"]), rtol=2e-6, atol=2e-6)
    for leaf, want in b8["probe"].items():
        np.testing.assert_allclose(np.asarray(a8["probe"][leaf]), np.asarray(want), rtol=0, atol=3e-6, err_msg=f"N=8 {leaf}")
    a, b = run(), run(IDQN_IQN_GEMM="0")
    assert a["losses"][0] == b["losses"][0]
    np.testing.assert_allclose(np.asarray(a["losses"]), np.asarray(b["losses"]), rtol=2e-6, atol=2e-6)
    for leaf, want in b["probe"].items():
        np.testing.assert_allclose(np.asarray(a["probe"][leaf]), np.asarray(want), rtol=0, atol=3e-6, err_msg=leaf)
    # the other launch-structure switches of the quantile heads: the embedding on the f32 MFMA instead of the pre-split bf16
    # planes, the two gradient GEMMs as two launches, other fraction groupings of the embedding / dL/dh kernels
    for env in ({"IDQN_IQN_EMBED3": "0"}, {"IDQN_IQN_MERGE": "0"},
                {"IDQN_IQN_EMBED_Q": "2", "IDQN_IQN_EMBED_QG": "8", "IDQN_IQN_DH_GROUPS": "2"}):
        c = run(**env)
        np.testing.assert_allclose(np.asarray(c["losses"]), np.asarray(a["losses"]), rtol=2e-6, atol=2e-6, err_msg=str(env))
        for leaf, want in a["probe"].items():
            np.testing.assert_allclose(np.asarray(c["probe"][leaf]), np.asarray(want), rtol=0, atol=3e-6, err_msg=f"{env} {leaf}")
