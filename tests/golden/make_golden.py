#!/usr/bin/env python
"""Generates tests/golden/*.npz from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference holds no golden vectors for CDAE and cannot be built or imported here (C++ needing
Eigen/Boost/glog/gflags), so these vectors pin the ORACLE — its line-by-line restatement of
/root/reference/src/model/recsys/cdae.hpp — against regressions, and give the HIP path fixed inputs and
expected outputs that do not depend on the oracle's code at test time.  Contents:
  step_kat.npz     one train_one_user_corruption per variant with explicit mask / negatives (incl. a
                   duplicate negative): z, y, g, hg and every parameter after the step
  loss_curve.npz   5 epochs of the literal schedule on a 60 x 100 dataset: loss per epoch, final params,
                   top-10 lists, Recall@10 / MAP@10
  loss_kat.npz     (pred, truth) -> (evaluate, gradient) grids for SQUARE and CROSS_ENTROPY incl. the +-18 branches
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cdae_amd import synth  # noqa: E402
import oracle as orc  # noqa: E402
from oracle import binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

VARIANTS = {
    "sq_tied_ada": dict(loss_type=ob.LOSS_SQUARE),
    "ce_tied_ada": dict(loss_type=ob.LOSS_CE),
    "ce_asym_ada": dict(loss_type=ob.LOSS_CE, asymmetric=True),
    "sq_asym_sgd": dict(loss_type=ob.LOSS_SQUARE, asymmetric=True, using_adagrad=False, learn_rate=0.02),
    "ce_tied_sgd_unscaled": dict(loss_type=ob.LOSS_CE, using_adagrad=False, learn_rate=0.02, scaled=False),
    "ce_tanh_nouser": dict(loss_type=ob.LOSS_CE, tanh=True, user_factor=False),
    "ce_tied_ada_gate": dict(loss_type=ob.LOSS_CE, linear_function=True),
    "sq_asym_ada_gate": dict(loss_type=ob.LOSS_SQUARE, asymmetric=True, linear_function=True),
}


def step_kat():
    U, I, K = 3, 32, 8
    rng = np.random.default_rng(20141119)
    rows = [np.sort(rng.choice(I, size=n, replace=False)).astype(np.uint32) for n in (6, 9, 4)]
    ptr = np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.int64)
    col = np.concatenate(rows)
    out = dict(U=U, I=I, K=K, ptr=ptr, col=col)
    # fixed fp32-representable parameters
    P = {ob.P_W: rng.uniform(-0.5, 0.5, (I, K)), ob.P_V: rng.uniform(-0.5, 0.5, (I, K)),
         ob.P_WU: rng.uniform(-0.5, 0.5, (U, K)), ob.P_B: rng.uniform(-0.2, 0.2, K),
         ob.P_BP: rng.uniform(-0.2, 0.2, I), ob.P_W_AG: rng.uniform(0.01, 0.5, (I, K)),
         ob.P_V_AG: rng.uniform(0.01, 0.5, (I, K)), ob.P_WU_AG: rng.uniform(0.01, 0.5, (U, K)),
         ob.P_B_AG: rng.uniform(0.01, 0.5, K), ob.P_BP_AG: rng.uniform(0.01, 0.5, I)}
    # drawn after everything else so that the older entries of the file keep their values
    P[ob.P_UU] = rng.uniform(0.5, 1.5, (U, K)); P[ob.P_UU_AG] = rng.uniform(0.01, 0.5, (U, K))
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    for k, v in P.items():
        out[f"init_{k}"] = v
    uid = 1
    kept = rows[uid][[0, 2, 3, 7]]
    free = np.setdiff1d(np.arange(I, dtype=np.uint32), rows[uid])
    neg = np.array([free[3], free[0], free[9], free[3], free[5]], dtype=np.uint32)     # free[3] twice
    out.update(uid=uid, kept=kept, neg=neg)
    for name, kw in VARIANTS.items():
        cfg = orc.OracleConfig(num_dim=K, num_neg=5, lambda_=0.01, learn_rate=kw.pop("learn_rate", 0.1),
                               corruption_ratio=0.5, beta=1.0, **kw)
        o = orc.Oracle(cfg, U, I, ptr, col)
        o.init_params(0)
        for k, v in P.items():
            if o.get(k).size:
                o.set(k, v)
        z, y, g, hg = o.step_user(uid, kept, neg)
        out[f"{name}_z"], out[f"{name}_y"], out[f"{name}_g"], out[f"{name}_hg"] = z, y, g, hg
        for k in P:
            if o.get(k).size:
                out[f"{name}_after_{k}"] = o.get(k)
    np.savez_compressed(os.path.join(OUT, "step_kat.npz"), **out)


def loss_curve():
    d = synth.generate(60, 100, 1800, seed=11, min_items=8)
    K, seed = 8, 20141119
    out = dict(train_ptr=d.train_ptr, train_col=d.train_col, test_ptr=d.test_ptr, test_col=d.test_col, K=K, seed=seed)
    for name, lt in (("sq", ob.LOSS_SQUARE), ("ce", ob.LOSS_CE)):
        cfg = orc.OracleConfig(num_dim=K, loss_type=lt, beta=1.0)
        o = orc.Oracle(cfg, d.num_users, d.num_items, d.train_ptr, d.train_col)
        o.init_params(seed)
        for k in range(10):       # start from fp32-representable parameters
            if o.get(k).size:
                o.set(k, o.get(k).astype(np.float32).astype(np.float64))
                out[f"{name}_init_{k}"] = o.get(k)
        losses = []
        for ep in range(5):
            o.train_literal(seed, ep)
            losses.append(o.data_loss(seed, ep) + o.penalty_loss())
        rec, sc = o.recommend(10, with_scores=True)
        out[f"{name}_loss"] = np.array(losses)
        out[f"{name}_rec"], out[f"{name}_rec_scores"] = rec, sc
        out[f"{name}_metrics"] = orc.eval_topn(rec, d.test_ptr, d.test_col)
        for k in range(10):
            if o.get(k).size:
                out[f"{name}_final_{k}"] = o.get(k)
    np.savez_compressed(os.path.join(OUT, "loss_curve.npz"), **out)


def loss_kat():
    d = synth.generate_shape("tiny", seed=5)
    preds = np.concatenate([np.linspace(-30, 30, 61), [-18.0000001, -17.9999999, 17.9999999, 18.0000001, 0.0]])
    out = dict(pred=preds)
    for name, lt in (("sq", ob.LOSS_SQUARE), ("ce", ob.LOSS_CE)):
        o = orc.Oracle(orc.OracleConfig(loss_type=lt), d.num_users, d.num_items, d.train_ptr, d.train_col)
        for t in (0, 1):
            out[f"{name}_eval_t{t}"] = np.array([o.loss_eval(p, float(t)) for p in preds])
            out[f"{name}_grad_t{t}"] = np.array([o.loss_grad(p, float(t)) for p in preds])
    np.savez_compressed(os.path.join(OUT, "loss_kat.npz"), **out)


if __name__ == "__main__":
    step_kat()
    loss_curve()
    loss_kat()
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
