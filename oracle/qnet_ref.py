"""CPU oracle (test infrastructure only): the i-DQN gradient step in numpy.

Restates, for ``architecture_type in {"cnn", "fc"}`` (and, forward only, ``"impala"`` -- the architecture the reference's
own unit tests use, tests/test_idqn.py:33; the HIP path has no impala kernels):

* ``DQNNet.__call__`` (``slimdqn/networks/architectures/dqn.py:38-70``): ``x / 255`` (cnn only),
  three ``flax.linen.Conv`` with the library default ``padding="SAME"`` (8x8/4, 4x4/2, 3x3/1;
  NHWC input, HWIO kernel, cross-correlation, bias added after), ReLU, flatten in (H, W, C) order,
  ``Dense + ReLU`` for ``features[3:]`` (cnn) or every feature (fc), final ``Dense(n_actions)``.
  SAME padding: ``out = ceil(i / s)``, ``p = max((out - 1) * s + k - i, 0)``, ``lo = p // 2``,
  ``hi = p - lo``  ->  84 -> 21 -> 11 -> 11 with pads (2,2), (1,2), (1,1); flatten = 7744.
* ``iDQN.compute_target`` / ``loss`` / ``loss_on_batch`` (``slimdqn/networks/idqn.py:111-124``):
  ``target = r + (1 - terminal) * gamma**n * max_a Q_target(s')``; ``loss = mean_b (Q(s)[a] - target)^2``.
* ``iDQN.learn_on_batch`` (``idqn.py:96-109``): per head ``value_and_grad`` w.r.t. the online
  parameters only, then ``optax.adam(lr, b1=0.9, b2=0.999, eps, eps_root=0)``:
  ``m = (1-b1) g + b1 m``; ``v = (1-b2) g^2 + b2 v``; ``t = count + 1``;
  ``theta += -lr * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps)``.
* ``shift_params`` / ``sync_target_params`` / the T-step copy (``idqn.py:13-24,74-94``).

The analytic backward here is hand-derived (im2col / col2im); ``oracle/torch_ref.py`` restates
the same step through torch autograd and ``tests/test_oracle_fp.py`` requires the two to agree.
jax/flax/optax are not installed, so nothing here is reference-captured: PARITY UNPINNED for the
fp32 path (see ``oracle/__init__.py``).

Parameters are flat dicts ``{"Conv_0/kernel": [K, 8, 8, C, F0], "Conv_0/bias": [K, F0], ...}`` with
the flax leaf names and a leading head axis K on every leaf (``idqn.py:48-50``).
"""
import numpy as np

B1, B2 = 0.9, 0.999


# ----------------------------------------------------------------------------------------------
# layout / init
# ----------------------------------------------------------------------------------------------
def same_pad(i, k, s):
    out = -(-i // s)
    p = max((out - 1) * s + k - i, 0)
    return out, p // 2, p - p // 2


CNN_GEOM = ((8, 4), (4, 2), (3, 1))  # (kernel, stride) of Conv_0..2, architectures/dqn.py:43-51


def leaf_shapes(arch, obs_dim, n_actions, features):
    """Ordered list of (name, per-head shape) in flax creation order."""
    out = []
    if arch == "cnn":
        h, w, c = obs_dim
        for li, (k, s) in enumerate(CNN_GEOM):
            out.append((f"Conv_{li}/kernel", (k, k, c, features[li])))
            out.append((f"Conv_{li}/bias", (features[li],)))
            h, w, c = same_pad(h, k, s)[0], same_pad(w, k, s)[0], features[li]
        fan, start = h * w * c, 3
    elif arch == "fc":
        fan = int(np.prod(obs_dim)) if not isinstance(obs_dim, (int, np.integer)) else int(obs_dim)
        start = 0
    elif arch == "impala":
        # architectures/dqn.py:7-29,54-60: three Stacks (Conv 3x3 SAME -> max_pool 3x3 / 2 SAME -> two residual blocks of
        # relu, Conv, relu, Conv), ReLU after the third, flatten; flax numbers a Stack's five convs Conv_0..Conv_4
        h, w, c = obs_dim
        for si in range(3):
            for ci in range(5):
                out.append((f"Stack_{si}/Conv_{ci}/kernel", (3, 3, c if ci == 0 else features[si], features[si])))
                out.append((f"Stack_{si}/Conv_{ci}/bias", (features[si],)))
            h, w, c = -(-h // 2), -(-w // 2), features[si]
        fan, start = h * w * c, 3
    else:
        raise NotImplementedError(arch)
    di = 0
    for f in list(features[start:]) + [n_actions]:
        out.append((f"Dense_{di}/kernel", (fan, f)))
        out.append((f"Dense_{di}/bias", (f,)))
        fan, di = f, di + 1
    return out


def init_params(seed, arch, obs_dim, n_actions, features, n_heads, dtype=np.float32):
    """Glorot-uniform kernels (cnn) / lecun-normal (fc), zero biases -- the reference's initialiser
    families (architectures/dqn.py:40,62); the jax PRNG stream itself is not reproduced."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape in leaf_shapes(arch, obs_dim, n_actions, features):
        if name.endswith("bias"):
            params[name] = np.zeros((n_heads,) + shape, dtype)
            continue
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in, fan_out = rf * shape[-2], rf * shape[-1]
        if arch == "cnn" or (arch == "impala" and ("Conv_0/" in name or name.startswith("Dense"))):
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            params[name] = rng.uniform(-lim, lim, size=(n_heads,) + shape).astype(dtype)
        else:
            std = np.sqrt(1.0 / fan_in) / 0.87962566103423978
            params[name] = np.clip(rng.standard_normal((n_heads,) + shape), -2, 2).astype(dtype) * dtype(std)
    return params


def head(params, k):
    return {n: a[k] for n, a in params.items()}


# ----------------------------------------------------------------------------------------------
# conv / dense primitives (single head)
# ----------------------------------------------------------------------------------------------
def _cols(x, k, s):
    b, h, w, c = x.shape
    oh, plo_h, phi_h = same_pad(h, k, s)
    ow, plo_w, phi_w = same_pad(w, k, s)
    xp = np.pad(x, ((0, 0), (plo_h, phi_h), (plo_w, phi_w), (0, 0)))
    win = np.lib.stride_tricks.sliding_window_view(xp, (k, k), axis=(1, 2))[:, ::s, ::s]  # B,OH,OW,C,kh,kw
    win = win[:, :oh, :ow].transpose(0, 1, 2, 4, 5, 3)  # B,OH,OW,kh,kw,C
    return np.ascontiguousarray(win).reshape(b * oh * ow, k * k * c), (oh, ow, plo_h, plo_w, xp.shape)


def conv_fwd(x, w, bias, s):
    k = w.shape[0]
    cols, (oh, ow, *_rest) = _cols(x, k, s)
    y = cols @ w.reshape(-1, w.shape[-1]) + bias
    return y.reshape(x.shape[0], oh, ow, -1), cols


def conv_bwd(x_shape, cols, w, s, dy, need_dx=True):
    k, co = w.shape[0], w.shape[-1]
    b, h, wd, c = x_shape
    dy2 = dy.reshape(-1, co)
    dw = (cols.T @ dy2).reshape(w.shape)
    db = dy2.sum(0)
    if not need_dx:
        return None, dw, db
    oh, plo_h, phi_h = same_pad(h, k, s)
    ow, plo_w, phi_w = same_pad(wd, k, s)
    dcols = (dy2 @ w.reshape(-1, co).T).reshape(b, oh, ow, k, k, c)
    dxp = np.zeros((b, h + plo_h + phi_h, wd + plo_w + phi_w, c), dy.dtype)
    for kh in range(k):
        for kw in range(k):
            dxp[:, kh : kh + oh * s : s, kw : kw + ow * s : s, :] += dcols[:, :, :, kh, kw, :]
    return dxp[:, plo_h : plo_h + h, plo_w : plo_w + wd, :], dw, db


def max_pool_same(x, k=3, s=2):
    """flax nn.max_pool(window (k, k), strides (s, s), padding="SAME"): -inf padding, NHWC."""
    b, h, w, c = x.shape
    oh, plo_h, phi_h = same_pad(h, k, s)
    ow, plo_w, phi_w = same_pad(w, k, s)
    xp = np.pad(x, ((0, 0), (plo_h, phi_h), (plo_w, phi_w), (0, 0)), constant_values=-np.inf)
    win = np.lib.stride_tricks.sliding_window_view(xp, (k, k), axis=(1, 2))[:, ::s, ::s][:, :oh, :ow]
    return win.max(axis=(-2, -1))


def impala_stack(ps, x, dtype=np.float64):
    """architectures/dqn.py:7-29 (ps: the Stack's leaves "Conv_i/kernel|bias")."""
    conv = lambda i, a: conv_fwd(a, ps[f"Conv_{i}/kernel"].astype(dtype), ps[f"Conv_{i}/bias"].astype(dtype), 1)[0]  # noqa: E731
    x = max_pool_same(conv(0, x))
    for blk in range(2):
        y = np.maximum(conv(1 + 2 * blk, np.maximum(x, 0)), 0)
        x = conv(2 + 2 * blk, y) + x
    return x


# ----------------------------------------------------------------------------------------------
# network forward / backward (single head)
# ----------------------------------------------------------------------------------------------
def forward(p, x, arch, dtype=np.float64, keep=False):
    """Q-values [B, A] of one head for a batch x ([B,H,W,C] uint8 for cnn, [B,...] float for fc)."""
    tape = []
    if arch == "cnn":
        a = x.astype(dtype) / dtype(255.0)
        for li, (k, s) in enumerate(CNN_GEOM):
            w, bias = p[f"Conv_{li}/kernel"].astype(dtype), p[f"Conv_{li}/bias"].astype(dtype)
            y, cols = conv_fwd(a, w, bias, s)
            out = np.maximum(y, 0)
            tape.append(("conv", li, a.shape, cols if keep else None, out))
            a = out
        a = a.reshape(a.shape[0], -1)
    elif arch == "impala":
        if keep:
            raise NotImplementedError("the impala restatement is forward-only (the reference's tests need no more; "
                                      "torch_ref differentiates it through autograd)")
        a = x.astype(dtype) / dtype(255.0)
        for si in range(3):
            a = impala_stack({n[len(f"Stack_{si}/"):]: v for n, v in p.items() if n.startswith(f"Stack_{si}/")}, a, dtype)
        a = np.maximum(a, 0)
        a = a.reshape(a.shape[0], -1)
    else:
        a = np.asarray(x).astype(dtype).reshape(x.shape[0], -1)
    n_dense = sum(1 for n in p if n.startswith("Dense_") and n.endswith("kernel"))
    for di in range(n_dense):
        w, bias = p[f"Dense_{di}/kernel"].astype(dtype), p[f"Dense_{di}/bias"].astype(dtype)
        y = a @ w + bias
        out = y if di == n_dense - 1 else np.maximum(y, 0)
        tape.append(("dense", di, a, None, out))
        a = out
    return (a, tape) if keep else a


def backward(p, tape, dq, dtype=np.float64, trace=None):
    """Gradients of every leaf of one head given dL/dQ [B, A].  trace (optional dict) receives the
    gradient w.r.t. every layer's pre-activation ("d_dense{i}", "d_conv{i}")."""
    grads = {}
    d = dq
    n_dense = sum(1 for t in tape if t[0] == "dense")
    for kind, li, inp, cols, out in reversed(tape):
        if kind == "dense":
            if li != n_dense - 1:
                d = d * (out > 0)
            if trace is not None:
                trace[f"d_dense{li}"] = d
            w = p[f"Dense_{li}/kernel"].astype(dtype)
            grads[f"Dense_{li}/kernel"] = inp.T @ d
            grads[f"Dense_{li}/bias"] = d.sum(0)
            d = d @ w.T
        else:
            d = d.reshape(out.shape) * (out > 0)
            if trace is not None:
                trace[f"d_conv{li}"] = d
            w = p[f"Conv_{li}/kernel"].astype(dtype)
            k, s = CNN_GEOM[li]
            dx, dw, db = conv_bwd(inp, cols, w, s, d, need_dx=(li > 0))
            grads[f"Conv_{li}/kernel"], grads[f"Conv_{li}/bias"] = dw, db
            d = dx
    return grads


# ----------------------------------------------------------------------------------------------
# the step
# ----------------------------------------------------------------------------------------------
def td_target(q_next, reward, terminal, gamma_n, dtype=np.float64):
    """idqn.py:120-124."""
    return reward.astype(dtype) + (1 - terminal.astype(np.int64)).astype(dtype) * dtype(gamma_n) * q_next.max(1)


def loss_and_grads(p_online, p_target, batch, arch, gamma_n, dtype=np.float64, weights=None):
    """One head: (loss, grads, aux) -- idqn.py:105,111-118.  ``weights`` (per-sample loss weights, default none) is
    NOT in the reference: it restates the prioritized-replay extension's loss  mean_b w_b * td_b**2  (SURVEY 8f-4)."""
    state, action, reward, next_state, terminal = batch
    bsz = state.shape[0]
    q, tape = forward(p_online, state, arch, dtype, keep=True)
    q_next = forward(p_target, next_state, arch, dtype)
    tgt = td_target(q_next, reward, terminal, gamma_n, dtype)
    td = q[np.arange(bsz), action] - tgt
    w = np.ones(bsz, dtype) if weights is None else np.asarray(weights, dtype)
    loss = (w * td * td).mean()
    dq = np.zeros_like(q)
    dq[np.arange(bsz), action] = 2.0 * w * td / bsz
    trace = {}
    grads = backward(p_online, tape, dq, dtype, trace)
    return loss, grads, {"q": q, "q_next": q_next, "target": tgt, "td": td, "tape": tape, "trace": trace}


def adam_update(theta, g, m, v, count, lr, eps, dtype=np.float64):
    """optax 0.2.4 scale_by_adam + scale(-lr) + apply_updates; count is the pre-increment step."""
    t = int(count) + 1
    m = dtype(1 - B1) * g + dtype(B1) * m
    v = dtype(1 - B2) * (g * g) + dtype(B2) * v
    bc1 = dtype(1.0 - np.float64(dtype(B1)) ** t)
    bc2 = dtype(1.0 - np.float64(dtype(B2)) ** t)
    upd = (m / bc1) / (np.sqrt(v / bc2) + dtype(eps))
    return theta + dtype(-lr) * upd, m, v


def learn_on_batch(params, target_params, mu, nu, count, batch, arch, gamma_n, lr, eps, dtype=np.float64,
                   return_grads=False):
    """All K heads: returns (params, mu, nu, count, losses[, grads]) -- idqn.py:96-109."""
    n_heads = next(iter(params.values())).shape[0]
    new_p = {n: a.astype(dtype).copy() for n, a in params.items()}
    new_m = {n: a.astype(dtype).copy() for n, a in mu.items()}
    new_v = {n: a.astype(dtype).copy() for n, a in nu.items()}
    losses = np.zeros(n_heads, dtype)
    all_grads = {n: np.zeros(a.shape, dtype) for n, a in params.items()}
    for k in range(n_heads):
        loss, grads, _ = loss_and_grads(head(params, k), head(target_params, k), batch, arch, gamma_n, dtype)
        losses[k] = loss
        for n in params:
            all_grads[n][k] = grads[n]
            new_p[n][k], new_m[n][k], new_v[n][k] = adam_update(
                new_p[n][k], grads[n].astype(dtype), new_m[n][k], new_v[n][k], count[k], lr, eps, dtype
            )
    out = (new_p, new_m, new_v, np.asarray(count) + 1, losses)
    return out + (all_grads,) if return_grads else out


def shift_params(params):
    """params[k] <- params[k+1] for k < K-1 (idqn.py:13-17)."""
    out = {}
    for n, a in params.items():
        b = a.copy()
        b[:-1] = a[1:]
        out[n] = b
    return out


def sync_target_params(params, target_params):
    """target[k] <- params[k-1] for k >= 1 (idqn.py:20-24)."""
    out = {}
    for n, a in params.items():
        b = target_params[n].copy()
        b[1:] = a[:-1]
        out[n] = b
    return out


def synthetic_batch(seed, bsz, obs_dim, n_actions, arch):
    """SURVEY 8d / BASELINE.md synthetic inputs."""
    rng = np.random.default_rng(seed)
    if arch == "cnn":
        s = rng.integers(0, 256, size=(bsz,) + tuple(obs_dim), dtype=np.uint8)
        s2 = rng.integers(0, 256, size=(bsz,) + tuple(obs_dim), dtype=np.uint8)
    else:
        d = (obs_dim,) if isinstance(obs_dim, (int, np.integer)) else tuple(obs_dim)
        s = rng.standard_normal((bsz,) + d + (1,)).astype(np.float32)
        s2 = rng.standard_normal((bsz,) + d + (1,)).astype(np.float32)
    a = rng.integers(0, n_actions, size=bsz).astype(np.int32)
    r = rng.integers(-1, 2, size=bsz).astype(np.float32)
    term = (rng.random(bsz) < 0.01)
    return s, a, r, s2, term
