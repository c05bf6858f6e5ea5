"""The acting kernels against the oracle at real size: Q-values of `idqn_act_host` (the single-state latency path,
csrc/act_kernels.h for the cnn, k_fc_q1 for the MLP; the hipGraph + host-mailbox route the trainer uses) compared with
``network.apply(params[idx], state)`` for EVERY head of both parameter sets, and the greedy action with ``jnp.argmax``.
Reference: slimdqn/networks/idqn.py:126-131 (best_action), slimdqn/sample_collection/utils.py:8-21 (select_action).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _agent(name):
    from oracle import make_golden as G
    from slimdqn.networks.idqn import iDQN

    arch, obs, A, feats, K, B, steps = G.FP_CASES[name]
    p, pt, batches = G.fp_case_inputs(name)
    agent = iDQN(0, obs, A, K, feats, arch, 6.25e-5, 0.99, 1, 1, 10**9, 10**9, adam_eps=1.5e-4)
    agent._load_flat(agent._online, p)
    agent._load_flat(agent._target, pt)
    return agent, arch, A, K, p, pt, batches[0]


@pytest.mark.parametrize("name", ["cnn_atari_k5", "cnn_atari_a18_b64", "fc_lunar_k3"])
def test_act_host_q_values_every_head(name):
    from oracle import qnet_ref as Q

    agent, arch, A, K, p, pt, batch = _agent(name)
    states = batch[0]
    worst = 0.0
    for which, params in ((0, p), (1, pt)):
        for head in range(K):
            for i in (0, 3):
                s = states[i]
                host = np.asarray(s)  # a HOST state: the idqn_act_host route (pinned upload, graph replay, polled mailbox)
                act = int(agent._best_action(which, head, host))
                q = agent._q_out[0].cpu().numpy()[:A]
                want = Q.forward(Q.head(params, head), states[i : i + 1], arch)[0]
                worst = max(worst, float(np.abs(q - want).max()))
                assert np.abs(q - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), (name, which, head, q, want)
                assert act == int(np.argmax(want))
                # the device-state route (idqn_best_action) must give the same numbers
                import torch

                dev = torch.from_numpy(np.ascontiguousarray(host)).cuda()
                act_d = int(agent._best_action(which, head, dev).item())
                q_d = agent._q_out[0].cpu().numpy()[:A]
                assert act_d == act and np.array_equal(q_d, q)
    print(f"[{name}] worst |q - oracle| = {worst:.2e}")


def test_batched_q_values_match_single_state_path():
    """idqn_q_values over n <= 32 states (the training kernels) and the single-state path agree with the oracle alike."""
    from oracle import qnet_ref as Q

    agent, arch, A, K, p, pt, batch = _agent("cnn_atari_k5")
    states = batch[0][:8]
    for head in (0, K - 1):
        q = agent.q_values(agent.params, states, head).cpu().numpy()
        want = Q.forward(Q.head(p, head), states, arch)
        assert np.abs(q - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
