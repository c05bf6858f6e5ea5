"""Run-time switches of the SHIPPED library must not change results: the same seeded steps are run in child processes (a switch is
read once per process) and compared with the default.  Six switches, each a second way of issuing the same arithmetic:

  IDQN_STEP_GRAPH=1     the plain step replayed as a hipGraph                       -> bit-identical to the eager launches
  IDQN_ACT_POLL=0       acting result by copy + synchronisation                     -> the polled mailbox's greedy actions
  IDQN_CONV_PP=0        one work item per workgroup instead of the persistent conv kernel (csrc/convp_pp.hip) where a launch has
                        several items per CU (B = 256)                              -> same losses / parameters within fp32 round-off
                        (the items are cut differently, the sums inside a tile are the same)
  IDQN_FC_PAR=0         the MLP step as k_fc_step_mfma + k_adam instead of the one-launch kernel (csrc/fc_par_kernels.h)
                                                                                    -> same within fp32 round-off
  IDQN_LOOP_OVERLAP=0   the trainer loop without the replay bookkeeping under the acting launch   -> bit-identical parameters
  IDQN_STEP_GRAPH in the trainer loop (the launcher switches it on)                 -> bit-identical parameters
(IDQN_CONV=f32, the f32-MFMA conv kernels, is a parity dimension of its own: every fp test runs in both modes; the fused
update_online_params call against its two halves is tests/test_gpu_int_path.py::test_learn_on_replay_is_sample_then_learn.)
The launch structures of rounds 2-5 that lost every A/B (chained convs, grouped / DMA-fed Dense_0 forward, stream roles, Adam
role, last-arriver finalize, full-row update, ...) were deleted in round 6 together with their switches; profiles/README.md
indexes their measurements.
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




def _run_child(code, **env):
    e = dict(os.environ)
    e.update(env)
    e.pop("IDQN_HIP_LIB", None)  # the shipped library
    out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + code], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):])


def _run(**env):
    return _run_child(CHILD, **env)


@pytest.fixture(scope="module")
def default_run():
    return _run()


def _close(a, b, rtol):
    for name in a["probe"]:
        np.testing.assert_allclose(np.asarray(a["probe"][name]), np.asarray(b["probe"][name]), rtol=rtol, atol=rtol * 1e-2, err_msg=name)
    np.testing.assert_allclose(np.asarray(a["losses"]), np.asarray(b["losses"]), rtol=rtol)


def test_step_graph_replay_is_bit_identical(default_run):
    got = _run(IDQN_STEP_GRAPH="1")
    assert got["losses"] == default_run["losses"]
    assert got["probe"] == default_run["probe"]


def test_acting_by_copy_and_sync_matches(default_run):
    got = _run(IDQN_ACT_POLL="0")
    assert got["acts"] == default_run["acts"]
    assert got["losses"] == default_run["losses"]


def test_persistent_conv_kernel_against_one_item_per_workgroup():
    """B = 256: the forward / data-gradient conv launches have several items per CU and run the persistent three-group kernel
    (csrc/convp_pp.hip); IDQN_CONV_PP=0 runs them as one item per workgroup.  The same 32 x 32 tiles with the same k order: the
    activations are bit-identical; the weight gradients' position chunks are untouched, so is everything else."""
    a, b = _run(SW_BATCH="256", SW_STEPS="4"), _run(SW_BATCH="256", SW_STEPS="4", IDQN_CONV_PP="0")
    assert a["losses"] == b["losses"]
    assert a["probe"] == b["probe"]
    assert a["acts"] == b["acts"]


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




def _trainer(**env):
    return _run_child(TRAINER_CHILD, **env)


def test_trainer_loop_with_and_without_step_graph():
    """The trainer's launcher switches IDQN_STEP_GRAPH on (experiments/base/launch.py).  The whole loop -- acting graphs,
    replay sampling fused into the step, T-step copies and shifts, D-step syncs in between -- must end with bit-identical
    online and target parameters whether the step is replayed as a graph or issued launch by launch."""
    a, b = _trainer(IDQN_STEP_GRAPH="1"), _trainer(IDQN_STEP_GRAPH="0")
    assert a["count"] == b["count"] and a["count"] >= 40
    assert a["probe"] == b["probe"]
    assert a["target"] == b["target"]


def test_trainer_loop_with_and_without_overlapped_replay_add():
    """The launcher lets the greedy action's launch return at once (idqn_act_host_begin / _end) and runs the replay
    bookkeeping of the previous transition under it (ReplayBuffer.add_deferred; IDQN_LOOP_OVERLAP=0 switches both off).  Same
    transitions in the same order, same samples, same steps: bit-identical parameters at the end."""
    a, b = _trainer(IDQN_LOOP_OVERLAP="1"), _trainer(IDQN_LOOP_OVERLAP="0")
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




def test_mlp_one_launch_kernel_against_the_two_launch_path():
    """The MLP step as ONE launch (k_fc_step_par: both forwards side by side, gradient assembled in LDS, coalesced Adam) against
    k_fc_step_mfma + k_adam (IDQN_FC_PAR=0): same tile arithmetic, other summation orders in the bias sums and Adam's
    reciprocal -> within fp32 round-off.  The [200, 200] net does not fit the one-launch plan and runs the same kernels either way."""
    a, b = _run_child(FC_CHILD), _run_child(FC_CHILD, IDQN_FC_PAR="0")
    for name in ("lunar", "wide"):
        _close(a[name], b[name], 2e-5)
    assert a["wide"] == b["wide"]
