"""Generates the committed golden vectors under tests/golden/ (run in the BUILD container only).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden [--int] [--fp]

--int  imports the reference's ``slimdqn/sample_collection/sum_tree.py`` as-is and its
       ``samplers.py`` (whose only obstacle is an unused ``import jax`` at samplers.py:7; an empty
       placeholder module object is put in ``sys.modules`` for that name -- no jax functionality is
       stood in for, nothing in samplers.py ever touches the name) from ``/root/reference`` and
       records operation traces: inputs AND the reference's outputs (node arrays, roots, query
       results, sampled keys, index maps).  Only data is written; no reference source travels.
       -> tests/golden/int_path_sumtree.npz, int_path_samplers.npz
--iqn  the same for the i-IQN extension (``oracle/iqn_ref.py``; the reference has no quantile code at all).
       -> tests/golden/fp_path_iqn_*.json
--fp   writes expected losses / gradient / post-Adam probes of the fp64 numpy restatement
       (``oracle/qnet_ref.py``) for seeded inputs.  NOT reference-captured (jax is not installed):
       "parity unpinned" for this part, see oracle/__init__.py.
       -> tests/golden/fp_path_*.json

The GPU box never has /root/reference: tests only read the files written here.
"""
import argparse
import hashlib
import json
import os
import sys
import types

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
REFERENCE = "/root/reference"


# ----------------------------------------------------------------------------------------------
# integer / fp64 path: captured from the reference
# ----------------------------------------------------------------------------------------------
def sumtree_op_script(seed, capacity, n_ops, max_batch):
    """A seeded script of set() calls: duplicates, float32-typed values, zeros, scalars."""
    rng = np.random.default_rng(seed)
    ops = []
    for j in range(n_ops):
        kind = j % 5
        if kind == 4:  # scalar form
            ops.append((int(rng.integers(capacity)), float(rng.random() * 3)))
            continue
        n = int(rng.integers(1, max_batch + 1))
        idx = rng.integers(capacity, size=n).astype(np.int32)
        if kind == 1 and n > 2:  # force duplicates with DIFFERENT values: first occurrence must win
            idx[1:] = idx[: n - 1]
        val = rng.random(n) * (10.0 ** rng.integers(-3, 3))
        if kind == 2:
            val = val.astype(np.float32)
        if kind == 3:
            val[rng.random(n) < 0.3] = 0.0
        ops.append((idx, val))
    return ops


def capture_sumtree():
    from slimdqn.sample_collection import sum_tree  # the reference, imported as-is

    out = {}
    cases = [(1, 6, 1), (4, 12, 4), (8, 16, 8), (100, 60, 32), (1000, 80, 64), (1 << 20, 40, 256), (3000, 30, 1024)]
    meta = []
    for ci, (cap, n_ops, max_batch) in enumerate(cases):
        tree = sum_tree.SumTree(cap)
        ops = sumtree_op_script(100 + ci, cap, n_ops, max_batch)
        roots, maxp, digests = [], [], []
        for j, (idx, val) in enumerate(ops):
            tree.set(idx, val)
            out[f"c{ci}_op{j}_idx"] = np.asarray(idx)
            out[f"c{ci}_op{j}_val"] = np.asarray(val)
            roots.append(tree.root)
            maxp.append(tree.max_recorded_priority)
            digests.append(hashlib.sha256(tree._nodes.tobytes()).hexdigest())
        rng = np.random.default_rng(7 + ci)
        targets = rng.uniform(0.0, tree.root, size=257)
        targets[0] = 0.0
        targets[1] = tree.root * (1.0 - 1e-6)
        out[f"c{ci}_roots"] = np.asarray(roots, np.float64)
        out[f"c{ci}_maxp"] = np.asarray(maxp, np.float64)
        out[f"c{ci}_q_targets"] = targets
        out[f"c{ci}_q_out"] = tree.query(targets)
        if cap <= 3000:
            out[f"c{ci}_nodes"] = tree._nodes.copy()
        probe = rng.integers(tree._nodes.size, size=64)
        out[f"c{ci}_probe_idx"] = probe
        out[f"c{ci}_probe_val"] = tree._nodes[probe]
        meta.append({"capacity": cap, "n_ops": len(ops), "depth": tree._depth, "n_nodes": int(tree._nodes.size),
                     "first_leaf": int(tree._first_leaf_offset), "digests": digests})
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLDEN, "int_path_sumtree.npz"), **out)
    print("wrote int_path_sumtree.npz", len(out), "arrays")


def capture_samplers():
    sys.modules.setdefault("jax", types.ModuleType("jax"))  # unused import at samplers.py:7
    from slimdqn.sample_collection import samplers  # the reference

    out, meta = {}, []
    # uniform: interleaved add / remove / sample; op codes 0=add 1=remove 2=sample(n)
    for ci, (seed, n_ops) in enumerate([(0, 200), (1, 400), (12345, 300)]):
        rng = np.random.default_rng(900 + ci)
        u = samplers.UniformSamplingDistribution(seed)
        live, next_key, script, samples = [], 0, [], []
        for j in range(n_ops):
            c = rng.random()
            if c < 0.5 or len(live) < 3:
                u.add(next_key); live.append(next_key); script.append((0, next_key)); next_key += 1
            elif c < 0.7:
                k = live.pop(int(rng.integers(len(live)))); u.remove(k); script.append((1, k))
            else:
                n = int(rng.integers(1, 40)); s = u.sample(n); script.append((2, n)); samples.append(s)
        out[f"u{ci}_script"] = np.asarray(script, np.int64)
        out[f"u{ci}_samples"] = np.concatenate(samples)
        out[f"u{ci}_final_index_to_key"] = np.asarray(u._index_to_key, np.int64)
        meta.append({"kind": "uniform", "seed": seed})
    # prioritized: 0=add(key,prio) 1=remove(key) 2=sample(n) 3=update(keys,prios)
    for ci, (seed, cap, alpha, n_ops) in enumerate([(0, 10, 1.0, 120), (3, 64, 0.6, 300), (5, 1000, 0.5, 400)]):
        rng = np.random.default_rng(700 + ci)
        p = samplers.PrioritizedSamplingDistribution(seed, cap, alpha)
        live, next_key, recs = [], 0, []
        for j in range(n_ops):
            c = rng.random()
            if (c < 0.45 or len(live) < 3) and len(live) < cap:
                pr = 0.0 if rng.random() < 0.15 else float(rng.random() * 5)
                p.add(next_key, priority=pr); live.append(next_key)
                recs.append({"op": 0, "key": next_key, "prio": pr}); next_key += 1
            elif c < 0.6 and len(live) > 3:
                k = live.pop(int(rng.integers(len(live)))); p.remove(k); recs.append({"op": 1, "key": k})
            elif c < 0.8:
                n = int(rng.integers(1, min(len(live), 16) + 1))
                ks = rng.choice(np.asarray(live), size=n, replace=bool(rng.random() < 0.3)).astype(np.int32)
                pr = rng.random(n) * 4
                pr[rng.random(n) < 0.2] = 0.0
                p.update(ks, pr); recs.append({"op": 3, "keys": ks.tolist(), "prios": pr.tolist()})
            else:
                if p._sum_tree.root == 0.0:
                    continue
                n = int(rng.integers(1, 40)); s = p.sample(n)
                recs.append({"op": 2, "n": n, "out": s.tolist(), "root": float(p._sum_tree.root)})
        meta.append({"kind": "prioritized", "seed": seed, "cap": cap, "alpha": alpha, "recs": recs,
                     "final_index_to_key": [int(k) for k in p._index_to_key],
                     "final_digest": hashlib.sha256(p._sum_tree._nodes.tobytes()).hexdigest()})
        out[f"p{ci}_final_nodes"] = p._sum_tree._nodes.copy()
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLDEN, "int_path_samplers.npz"), **out)
    print("wrote int_path_samplers.npz")


# ----------------------------------------------------------------------------------------------
# fp path: from the fp64 restatement (not reference-captured)
# ----------------------------------------------------------------------------------------------
FP_CASES = {
    # name: (arch, obs_dim, n_actions, features, K, B, steps)
    "cnn_small": ("cnn", (20, 20, 4), 5, [32, 32, 32, 128], 2, 32, 2),
    "cnn_atari_k5": ("cnn", (84, 84, 4), 6, [32, 64, 64, 512], 5, 32, 2),
    "cnn_atari_a18_b64": ("cnn", (84, 84, 4), 18, [32, 64, 64, 512], 2, 64, 1),
    "fc_lunar_k3": ("fc", 8, 4, [100, 100], 3, 32, 3),
    # BASELINE configs 4 and 5 at their full single-device workload (one step each; probe indices keep the fixtures small)
    "cnn_atari_k5_b256": ("cnn", (84, 84, 4), 6, [32, 64, 64, 512], 5, 256, 1),
    "cnn_atari_k64": ("cnn", (84, 84, 4), 6, [32, 64, 64, 512], 64, 32, 1),
}
FP_HYPER = {"gamma": 0.99, "n": 1, "lr": 6.25e-5, "eps": 1.5e-4}


def fp_case_inputs(name):
    from . import qnet_ref as Q

    arch, obs, A, feats, K, B, steps = FP_CASES[name]
    seed = sum(name.encode())
    p = Q.init_params(seed, arch, obs, A, feats, K, np.float32)
    pt = Q.init_params(seed + 1, arch, obs, A, feats, K, np.float32)
    rng = np.random.default_rng(seed + 2)
    for n in p:  # non-zero biases so that bias handling is exercised
        if n.endswith("bias"):
            p[n] = (0.05 * rng.standard_normal(p[n].shape)).astype(np.float32)
            pt[n] = (0.05 * rng.standard_normal(p[n].shape)).astype(np.float32)
    batches = []
    for s in range(steps):
        st, a, r, s2, term = Q.synthetic_batch(seed + 10 + s, B, obs, A, arch)
        term[s % B] = True
        batches.append((st, a, r, s2, term))
    return p, pt, batches


def probe_indices(name, leaf, size, n=24):
    rng = np.random.default_rng(sum((name + leaf).encode()))
    return np.sort(rng.choice(size, size=min(n, size), replace=False))


def capture_fp(names=None):
    from . import qnet_ref as Q

    for name in names or FP_CASES:
        arch, obs, A, feats, K, B, steps = FP_CASES[name]
        p, pt, batches = fp_case_inputs(name)
        mu = {n: np.zeros_like(a, dtype=np.float64) for n, a in p.items()}
        nu = {n: np.zeros_like(a, dtype=np.float64) for n, a in p.items()}
        p64 = {n: a.astype(np.float64) for n, a in p.items()}
        count = np.zeros(K, np.int64)
        rec = {"case": name, "hyper": FP_HYPER, "steps": []}
        gamma_n = FP_HYPER["gamma"] ** FP_HYPER["n"]
        for s, batch in enumerate(batches):
            # forward probes for head 0 (first step only): q, q_next
            if s == 0:
                _, _, aux = Q.loss_and_grads(Q.head(p64, 0), Q.head(pt, 0), batch, arch, gamma_n)
                rec["q_head0"] = aux["q"].tolist()
                rec["q_next_head0"] = aux["q_next"].tolist()
            p64, mu, nu, count, losses, grads = Q.learn_on_batch(
                p64, pt, mu, nu, count, batch, arch, gamma_n, FP_HYPER["lr"], FP_HYPER["eps"], np.float64,
                return_grads=True)
            step = {"losses": losses.tolist(), "leaves": {}}
            for leaf in p64:
                flat_g = grads[leaf].reshape(K, -1)
                flat_p = p64[leaf].reshape(K, -1)
                idx = probe_indices(name, leaf, flat_g.shape[1])
                step["leaves"][leaf] = {
                    "idx": idx.tolist(),
                    "grad": flat_g[:, idx].tolist(),
                    "param": flat_p[:, idx].tolist(),
                    "grad_l2": np.sqrt((flat_g**2).sum(1)).tolist(),
                    "grad_absmax": np.abs(flat_g).max(1).tolist(),
                }
            rec["steps"].append(step)
        with open(os.path.join(GOLDEN, f"fp_path_{name}.json"), "w") as f:
            json.dump(rec, f)
        print("wrote", name, [s["losses"] for s in rec["steps"]])


# ----------------------------------------------------------------------------------------------
# i-IQN extension (oracle/iqn_ref.py; no reference code exists for it): fp64 restatement -> probes
# ----------------------------------------------------------------------------------------------
IQN_CASES = {  # name: (obs, A, features, K, B, N)
    "iqn_small": ((20, 20, 4), 5, [32, 32, 32, 256], 2, 32, 4),
    "iqn_small_ragged": ((20, 20, 4), 5, [32, 64, 32, 256], 2, 20, 5),
    "iqn_atari_k5": ((84, 84, 4), 6, [32, 64, 64, 512], 5, 32, 32),
}


def iqn_case_inputs(name):
    from . import iqn_ref as I
    from . import qnet_ref as Q

    obs, A, feats, K, B, N = IQN_CASES[name]
    seed = sum(name.encode())
    p = I.init_params(seed, obs, A, feats, K)
    pt = I.init_params(seed + 1, obs, A, feats, K)
    rng = np.random.default_rng(seed + 2)
    for n in p:
        if n.endswith("bias"):
            p[n] = (0.05 * rng.standard_normal(p[n].shape)).astype(np.float32)
            pt[n] = (0.05 * rng.standard_normal(p[n].shape)).astype(np.float32)
    st, a, r, s2, term = Q.synthetic_batch(seed + 10, B, obs, A, "cnn")
    term[0] = True
    taus = I.synthetic_taus(seed + 20, K, N, B)
    return p, pt, (st, a, r, s2, term), taus


def capture_iqn(names=None):
    from . import iqn_ref as I
    from . import qnet_ref as Q

    for name in names or IQN_CASES:
        obs, A, feats, K, B, N = IQN_CASES[name]
        p, pt, batch, taus = iqn_case_inputs(name)
        gamma_n = FP_HYPER["gamma"] ** FP_HYPER["n"]
        mu = {n: np.zeros_like(a, dtype=np.float64) for n, a in p.items()}
        nu = {n: np.zeros_like(a, dtype=np.float64) for n, a in p.items()}
        _, _, aux0 = I.loss_and_grads(Q.head(p, 0), Q.head(pt, 0), batch, tuple(taus[0]), gamma_n)
        p64, mu, nu, count, losses, grads = I.learn_on_batch(p, pt, mu, nu, np.zeros(K, np.int64), batch, taus, gamma_n,
                                                             FP_HYPER["lr"], FP_HYPER["eps"], np.float64, return_grads=True)
        rec = {"case": name, "hyper": FP_HYPER, "losses": losses.tolist(), "z_online_head0": aux0["z_a"].tolist(),
               "z_target_head0": aux0["z_t"].tolist(), "a_star_head0": aux0["a_star"].tolist(),
               "q_select_head0": aux0["q_sel"].tolist(), "leaves": {}}
        for leaf in p64:
            flat_g, flat_p = grads[leaf].reshape(K, -1), p64[leaf].reshape(K, -1)
            idx = probe_indices(name, leaf, flat_g.shape[1])
            rec["leaves"][leaf] = {"idx": idx.tolist(), "grad": flat_g[:, idx].tolist(), "param": flat_p[:, idx].tolist(),
                                   "grad_l2": np.sqrt((flat_g**2).sum(1)).tolist(), "grad_absmax": np.abs(flat_g).max(1).tolist()}
        with open(os.path.join(GOLDEN, f"fp_path_{name}.json"), "w") as f:
            json.dump(rec, f)
        print("wrote", name, rec["losses"])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--int", action="store_true")
    ap.add_argument("--fp", action="store_true")
    ap.add_argument("--iqn", action="store_true")
    ap.add_argument("--cases", nargs="*")
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    if args.int:
        sys.path.insert(0, REFERENCE)
        capture_sumtree()
        capture_samplers()
    if args.fp:
        capture_fp(args.cases)
    if args.iqn:
        capture_iqn(args.cases)
