"""CPU oracle (test infrastructure only): the i-DQN gradient step through torch-CPU autograd.

An independent second restatement of the same reference code as ``oracle/qnet_ref.py``
(``slimdqn/networks/idqn.py:96-124``, ``slimdqn/networks/architectures/dqn.py:38-70``, optax adam):
the backward pass here comes from autograd rather than from hand-derived formulae, the conv is
``F.conv2d`` on NCHW/OIHW views with explicit asymmetric ``F.pad`` for flax's SAME rule.

Two uses:
* ``tests/test_oracle_fp.py`` requires it to agree with ``qnet_ref`` (fp64, < 1e-9 rel);
* ``bench.py``'s ``cpu_baseline`` leg times ``BatchedStep`` (K heads batched as one grouped conv /
  bmm, fp32, all host cores) as the "CPU restatement of the reference path" that BASELINE.md
  prescribes in place of the uninstallable JAX-CPU run (``cpu_baseline.kind = "port"``).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .qnet_ref import B1, B2, CNN_GEOM, same_pad


def _t(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype)


def forward_head(p, x, arch, dtype=torch.float64):
    """p: dict name -> torch tensor (single head). x: [B,H,W,C] uint8 (cnn) or [B,...] float (fc)."""
    if arch == "cnn":
        a = (x.to(dtype) / 255.0).permute(0, 3, 1, 2)
        for li, (k, s) in enumerate(CNN_GEOM):
            w = p[f"Conv_{li}/kernel"].permute(3, 2, 0, 1)  # HWIO -> OIHW
            _, lo_h, hi_h = same_pad(a.shape[2], k, s)
            _, lo_w, hi_w = same_pad(a.shape[3], k, s)
            a = F.relu(F.conv2d(F.pad(a, (lo_w, hi_w, lo_h, hi_h)), w, p[f"Conv_{li}/bias"], stride=s))
        a = a.permute(0, 2, 3, 1).reshape(a.shape[0], -1)
    elif arch == "impala":  # architectures/dqn.py:7-29,54-60
        a = (x.to(dtype) / 255.0).permute(0, 3, 1, 2)
        conv = lambda n, t: F.conv2d(F.pad(t, (1, 1, 1, 1)), p[n + "/kernel"].permute(3, 2, 0, 1), p[n + "/bias"])  # noqa: E731
        for si in range(3):
            a = conv(f"Stack_{si}/Conv_0", a)
            _, lo_h, hi_h = same_pad(a.shape[2], 3, 2)
            _, lo_w, hi_w = same_pad(a.shape[3], 3, 2)
            a = F.max_pool2d(F.pad(a, (lo_w, hi_w, lo_h, hi_h), value=float("-inf")), 3, 2)
            for blk in range(2):
                y = F.relu(conv(f"Stack_{si}/Conv_{1 + 2 * blk}", F.relu(a)))
                a = conv(f"Stack_{si}/Conv_{2 + 2 * blk}", y) + a
        a = F.relu(a).permute(0, 2, 3, 1).reshape(a.shape[0], -1)
    else:
        a = x.to(dtype).reshape(x.shape[0], -1)
    n_dense = sum(1 for n in p if n.startswith("Dense_") and n.endswith("kernel"))
    for di in range(n_dense):
        a = a @ p[f"Dense_{di}/kernel"] + p[f"Dense_{di}/bias"]
        if di != n_dense - 1:
            a = F.relu(a)
    return a


def loss_and_grads(p_online, p_target, batch, arch, gamma_n, dtype=torch.float64):
    state, action, reward, next_state, terminal = batch
    po = {n: _t(a, dtype).requires_grad_(True) for n, a in p_online.items()}
    pt = {n: _t(a, dtype) for n, a in p_target.items()}
    s, s2 = torch.as_tensor(state), torch.as_tensor(next_state)
    q = forward_head(po, s, arch, dtype)
    with torch.no_grad():
        qn = forward_head(pt, s2, arch, dtype)
    tgt = _t(reward, dtype) + (1 - _t(terminal.astype(np.int64), dtype)) * gamma_n * qn.max(1).values
    qa = q.gather(1, torch.as_tensor(action.astype(np.int64))[:, None])[:, 0]
    loss = ((qa - tgt) ** 2).mean()
    loss.backward()
    return float(loss.detach()), {n: t.grad.numpy() for n, t in po.items()}


class BatchedStep:
    """K heads at once, fp32, Nature-CNN: the timed CPU baseline (and a third cross-check)."""

    def __init__(self, params, target_params, n_actions, gamma_n, lr, eps, dtype=torch.float32):
        self.K = next(iter(params.values())).shape[0]
        self.dtype, self.gamma_n, self.lr, self.eps, self.A = dtype, gamma_n, lr, eps, n_actions
        self.p = {n: _t(a, dtype).requires_grad_(True) for n, a in params.items()}
        self.pt = {n: _t(a, dtype) for n, a in target_params.items()}
        self.m = {n: torch.zeros_like(t) for n, t in self.p.items()}
        self.v = {n: torch.zeros_like(t) for n, t in self.p.items()}
        self.count = 0

    def _fwd(self, p, x):
        K = self.K
        a = (x.to(self.dtype) / 255.0).permute(0, 3, 1, 2)  # B,C,H,W
        a = a.repeat(1, K, 1, 1)  # heads stacked on the channel axis, one group per head
        for li, (k, s) in enumerate(CNN_GEOM):
            w = p[f"Conv_{li}/kernel"]  # K,kh,kw,ci,co
            w = w.permute(0, 4, 3, 1, 2).reshape(K * w.shape[4], w.shape[3], k, k)
            _, lo_h, hi_h = same_pad(a.shape[2], k, s)
            _, lo_w, hi_w = same_pad(a.shape[3], k, s)
            a = F.relu(F.conv2d(F.pad(a, (lo_w, hi_w, lo_h, hi_h)), w, p[f"Conv_{li}/bias"].reshape(-1),
                                stride=s, groups=K))
        b, _, h, w_ = a.shape
        a = a.reshape(b, K, -1, h, w_).permute(1, 0, 3, 4, 2).reshape(K, b, -1)  # K,B,(H,W,C)
        a = F.relu(torch.baddbmm(p["Dense_0/bias"][:, None, :], a, p["Dense_0/kernel"]))
        return torch.baddbmm(p["Dense_1/bias"][:, None, :], a, p["Dense_1/kernel"])  # K,B,A

    def step(self, batch):
        state, action, reward, next_state, terminal = batch
        s, s2 = torch.as_tensor(state), torch.as_tensor(next_state)
        with torch.no_grad():
            qn = self._fwd(self.pt, s2)
            tgt = _t(reward, self.dtype) + (1 - _t(terminal.astype(np.int64), self.dtype)) * self.gamma_n * qn.max(2).values
        q = self._fwd(self.p, s)
        idx = torch.as_tensor(action.astype(np.int64))[None, :, None].expand(self.K, -1, 1)
        losses = ((q.gather(2, idx)[:, :, 0] - tgt) ** 2).mean(1)
        grads = torch.autograd.grad(losses.sum(), list(self.p.values()))
        self.count += 1
        bc1, bc2 = 1 - B1 ** self.count, 1 - B2 ** self.count
        with torch.no_grad():
            for (n, t), g in zip(self.p.items(), grads):
                self.m[n].mul_(B1).add_(g, alpha=1 - B1)
                self.v[n].mul_(B2).addcmul_(g, g, value=1 - B2)
                t.add_((self.m[n] / bc1) / ((self.v[n] / bc2).sqrt() + self.eps), alpha=-self.lr)
        return losses.detach().numpy()


# ----------------------------------------------------------------------------------------------
# i-IQN extension (oracle/iqn_ref.py states the algorithm and its sources): the same step through autograd
# ----------------------------------------------------------------------------------------------
def iqn_quantile_values(p, psi, tau, dtype=torch.float64):
    i = torch.arange(1, 65, dtype=torch.float64)
    c = torch.cos(np.pi * i * torch.as_tensor(tau).to(torch.float64)[..., None]).to(dtype)  # [N, B, 64]
    phi = F.relu(c @ p["Embed_0/kernel"] + p["Embed_0/bias"])
    h = F.relu((psi[None] * phi) @ p["Dense_0/kernel"] + p["Dense_0/bias"])
    return h @ p["Dense_1/kernel"] + p["Dense_1/bias"]


def iqn_loss_and_grads(p_online, p_target, batch, taus, gamma_n, dtype=torch.float64, kappa=1.0):
    state, action, reward, next_state, terminal = batch
    tau_on, tau_sel, tau_tg = taus
    po = {n: _t(a, dtype).requires_grad_(True) for n, a in p_online.items()}
    pt = {n: _t(a, dtype) for n, a in p_target.items()}
    conv = lambda p: {n: a for n, a in p.items() if n.startswith("Conv_")}  # noqa: E731
    bsz = state.shape[0]
    ar = torch.arange(bsz)
    z = iqn_quantile_values(po, forward_head(conv(po), torch.as_tensor(state), "cnn", dtype), tau_on, dtype)
    with torch.no_grad():
        psi_t = forward_head(conv(pt), torch.as_tensor(next_state), "cnn", dtype)
        a_star = iqn_quantile_values(pt, psi_t, tau_sel, dtype).mean(0).argmax(1)
        z_t = iqn_quantile_values(pt, psi_t, tau_tg, dtype)[:, ar, a_star]
        tgt = _t(reward, dtype)[None] + (1 - _t(terminal.astype(np.int64), dtype))[None] * gamma_n * z_t
    z_a = z[:, ar, torch.as_tensor(action.astype(np.int64))]
    delta = tgt[:, None, :] - z_a[None, :, :]
    hub = torch.where(delta.abs() <= kappa, 0.5 * delta**2, kappa * (delta.abs() - 0.5 * kappa))
    wgt = (torch.as_tensor(tau_on).to(dtype)[None] - (delta.detach() < 0).to(dtype)).abs()
    loss = (wgt * hub / kappa).sum(1).mean(0).mean()
    loss.backward()
    return float(loss.detach()), {n: t.grad.numpy() for n, t in po.items()}
