"""CPU oracle (test infrastructure only): the i-IQN gradient step in numpy.  EXTENSION -- PARITY UNPINNED.

BASELINE config 3 ("Atari i-IQN K=5, 32 quantile samples") has no counterpart in the reference snapshot: its README
names i-IQN and points at another repository (``/root/reference/README.md:3,10``), there is no quantile code, test or
vector.  What is restated here is therefore the PUBLISHED algorithm, combined with the reference's own chain of heads:

* heads / targets / Adam / shift / sync exactly as i-DQN (``slimdqn/networks/idqn.py:13-24,96-109``): head k regresses
  onto ``target_params[k]``, K independent heads inside a step;
* the implicit quantile network of Dabney et al. 2018 (IQN, arXiv:1806.06923, section 3 and Appendix), in the form of
  Dopamine's JAX ``ImplicitQuantileNetwork`` / ``JaxImplicitQuantileAgent``, on top of the reference's conv trunk
  (``slimdqn/networks/architectures/dqn.py:39-53``: x/255, three SAME convs + ReLU, (H, W, C) flatten):
      psi(s)    = trunk(s)                                             [F]
      phi(tau)  = relu(Embed_0(cos(pi * i * tau)), i = 1..64)          [F]      (Dense 64 -> F with bias)
      Z(s, tau) = Dense_1(relu(Dense_0(psi(s) * phi(tau))))            [A]      (Dense_0: F -> features[3], Dense_1: -> A)
* the loss of one sample with N online quantiles tau_j, N_sel action-selection quantiles and N' target quantiles tau'_i:
      a*      = argmax_a mean_l Z_target(s', tau~_l)[a]                (first maximum on ties, like jnp.argmax)
      t_i     = r + (1 - terminal) * gamma**n * Z_target(s', tau'_i)[a*]
      delta_ij = t_i - Z_online(s, tau_j)[a]
      rho_ij  = |tau_j - 1[delta_ij < 0]| * huber_kappa(delta_ij) / kappa,  kappa = 1   (indicator not differentiated)
      L       = mean_b (1 / N') sum_i sum_j rho_ij                     (sum over online, mean over target quantiles)
  and ``cos(pi i tau)`` evaluated exactly (the HIP path rounds the fp64 value to f32 once).
The quantile fractions are INPUTS (the host draws them from numpy's PCG64), so every comparison is deterministic.

Leaves: the reference's cnn leaves (``Conv_0..2``, ``Dense_0`` [F, J], ``Dense_1`` [J, A]) plus ``Embed_0/kernel``
[64, F] and ``Embed_0/bias`` [F] appended (naming is this build's: flax would number the three Dense layers in call
order).  ``oracle/torch_ref.iqn_loss_and_grads`` restates the same step through autograd; tests require agreement.
"""
import numpy as np

from . import qnet_ref as Q

EMBED_DIM = 64
KAPPA = 1.0


def leaf_shapes(obs_dim, n_actions, features):
    base = Q.leaf_shapes("cnn", obs_dim, n_actions, features)
    fan = dict(base)["Dense_0/kernel"][0]
    return base + [("Embed_0/kernel", (EMBED_DIM, fan)), ("Embed_0/bias", (fan,))]


def init_params(seed, obs_dim, n_actions, features, n_heads, dtype=np.float32):
    """Glorot-uniform kernels, zero biases (the reference's cnn initialiser family, architectures/dqn.py:40)."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape in leaf_shapes(obs_dim, n_actions, features):
        if name.endswith("bias"):
            params[name] = np.zeros((n_heads,) + shape, dtype)
            continue
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        lim = np.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
        params[name] = rng.uniform(-lim, lim, size=(n_heads,) + shape).astype(dtype)
    return params


def synthetic_taus(seed, n_heads, n_quantiles, bsz):
    """tau[K][3][N][B] in (0, 1): online, action-selection and target fractions of every head (float32 values)."""
    rng = np.random.default_rng(seed)
    return rng.random((n_heads, 3, n_quantiles, bsz)).astype(np.float32)


def cos_features(tau, dtype=np.float64):
    """[N, B] -> [N, B, 64]: cos(pi * i * tau), i = 1..64, from the float32 fraction, in fp64."""
    i = np.arange(1, EMBED_DIM + 1, dtype=np.float64)
    return np.cos(np.pi * i * tau.astype(np.float64)[..., None]).astype(dtype)


def trunk(p, x, dtype=np.float64, keep=False):
    conv = {n: a for n, a in p.items() if n.startswith("Conv_")}
    out, tape = Q.forward(conv, x, "cnn", dtype, keep=True)  # (no Dense_* leaves in `conv`: the flattened features)
    return (out, tape) if keep else out


def quantile_values(p, psi, tau, dtype=np.float64, keep=False):
    """Z [N, B, A] for features psi [B, F] and fractions tau [N, B]."""
    c = cos_features(tau, dtype)
    e = c @ p["Embed_0/kernel"].astype(dtype) + p["Embed_0/bias"].astype(dtype)
    phi = np.maximum(e, 0)
    x = psi[None] * phi
    pre = x @ p["Dense_0/kernel"].astype(dtype) + p["Dense_0/bias"].astype(dtype)
    h = np.maximum(pre, 0)
    z = h @ p["Dense_1/kernel"].astype(dtype) + p["Dense_1/bias"].astype(dtype)
    return (z, (c, e, phi, x, h)) if keep else z


def huber(d):
    a = np.abs(d)
    return np.where(a <= KAPPA, 0.5 * d * d, KAPPA * (a - 0.5 * KAPPA))


def huber_grad(d):
    return np.where(np.abs(d) <= KAPPA, d, KAPPA * np.sign(d))


def loss_and_grads(p_online, p_target, batch, taus, gamma_n, dtype=np.float64):
    """One head.  taus = (tau_online [N, B], tau_select [N_sel, B], tau_target [N', B])."""
    state, action, reward, next_state, terminal = batch
    tau_on, tau_sel, tau_tg = taus
    bsz = state.shape[0]
    n_on, n_tg = tau_on.shape[0], tau_tg.shape[0]
    psi, tape = trunk(p_online, state, dtype, keep=True)
    z, (c, e, phi, x, h) = quantile_values(p_online, psi, tau_on, dtype, keep=True)
    psi_t = trunk(p_target, next_state, dtype)
    q_sel = quantile_values(p_target, psi_t, tau_sel, dtype).mean(0)  # [B, A]
    a_star = q_sel.argmax(1)
    z_t = quantile_values(p_target, psi_t, tau_tg, dtype)[:, np.arange(bsz), a_star]  # [N', B]
    tgt = reward.astype(dtype)[None] + (1 - terminal.astype(np.int64)).astype(dtype)[None] * dtype(gamma_n) * z_t
    z_a = z[:, np.arange(bsz), action]  # [N, B]
    delta = tgt[:, None, :] - z_a[None, :, :]  # [N', N, B]
    wgt = np.abs(tau_on.astype(dtype)[None] - (delta < 0).astype(dtype))
    rho = wgt * huber(delta) / KAPPA
    per_sample = rho.sum(1).mean(0)  # sum over the online fractions, mean over the target ones
    loss = per_sample.mean()
    # dL/dz_a[j, b] = -(1 / (B N')) sum_i wgt_ij huber'(delta_ij) / kappa
    dz_a = -(wgt * huber_grad(delta) / KAPPA).sum(0) / (bsz * n_tg)
    dz = np.zeros_like(z)
    dz[:, np.arange(bsz), action] = dz_a
    grads = {}
    w1, w0 = p_online["Dense_1/kernel"].astype(dtype), p_online["Dense_0/kernel"].astype(dtype)
    grads["Dense_1/kernel"] = np.einsum("nbj,nba->ja", h, dz)
    grads["Dense_1/bias"] = dz.sum((0, 1))
    dh = (dz @ w1.T) * (h > 0)
    grads["Dense_0/kernel"] = np.einsum("nbf,nbj->fj", x, dh)
    grads["Dense_0/bias"] = dh.sum((0, 1))
    dx = dh @ w0.T  # [N, B, F]
    dphi = dx * psi[None] * (e > 0)
    grads["Embed_0/kernel"] = np.einsum("nbi,nbf->if", c, dphi)
    grads["Embed_0/bias"] = dphi.sum((0, 1))
    dpsi = (dx * phi).sum(0)  # [B, F]
    conv = {n: a for n, a in p_online.items() if n.startswith("Conv_")}
    trace = {}
    grads.update(Q.backward(conv, tape, dpsi, dtype, trace))
    aux = {"z": z, "z_a": z_a, "q_sel": q_sel, "a_star": a_star, "z_t": z_t, "target": tgt, "per_sample": per_sample,
           "psi": psi, "dh": dh, "dpsi": dpsi, "dz_a": dz_a, "trace": trace, "n_on": n_on}
    return loss, grads, aux


def learn_on_batch(params, target_params, mu, nu, count, batch, taus, gamma_n, lr, eps, dtype=np.float64, return_grads=False):
    """All K heads (independent inside a step, idqn.py:96-109); taus [K][3][N][B]."""
    n_heads = next(iter(params.values())).shape[0]
    new_p = {n: a.astype(dtype).copy() for n, a in params.items()}
    new_m = {n: a.astype(dtype).copy() for n, a in mu.items()}
    new_v = {n: a.astype(dtype).copy() for n, a in nu.items()}
    losses = np.zeros(n_heads, dtype)
    all_grads = {n: np.zeros(a.shape, dtype) for n, a in params.items()}
    for k in range(n_heads):
        loss, grads, _ = loss_and_grads(Q.head(params, k), Q.head(target_params, k), batch, tuple(taus[k]), gamma_n, dtype)
        losses[k] = loss
        for n in params:
            all_grads[n][k] = grads[n]
            new_p[n][k], new_m[n][k], new_v[n][k] = Q.adam_update(new_p[n][k], grads[n].astype(dtype), new_m[n][k],
                                                                  new_v[n][k], count[k], lr, eps, dtype)
    out = (new_p, new_m, new_v, np.asarray(count) + 1, losses)
    return out + (all_grads,) if return_grads else out


def greedy_action(p, state, tau, dtype=np.float64):
    """argmax_a mean_l Z(s, tau_l)[a] for ONE state (tau [N]): the acting rule of IQN."""
    psi = trunk(p, state[None], dtype)
    q = quantile_values(p, psi, tau[:, None], dtype).mean(0)[0]
    return int(q.argmax()), q
