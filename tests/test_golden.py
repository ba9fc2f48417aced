"""Golden vectors under tests/golden/ (made by tests/golden/make_golden.py from the oracle).

CPU tests: the oracle still reproduces them (regression pin of the restatement).
GPU tests: the HIP path reproduces them through the C ABI with no oracle code involved at test time.
"""
import os

import numpy as np
import pytest

import cdae_amd
import oracle as orc
from oracle import binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
G = lambda name: np.load(os.path.join(HERE, "golden", name))   # noqa: E731

VARIANTS = {
    "sq_tied_ada": dict(loss=ob.LOSS_SQUARE),
    "ce_tied_ada": dict(loss=ob.LOSS_CE),
    "ce_asym_ada": dict(loss=ob.LOSS_CE, asymmetric=True),
    "sq_asym_sgd": dict(loss=ob.LOSS_SQUARE, asymmetric=True, using_adagrad=False, learn_rate=0.02),
    "ce_tied_sgd_unscaled": dict(loss=ob.LOSS_CE, using_adagrad=False, learn_rate=0.02, scaled=False),
    "ce_tanh_nouser": dict(loss=ob.LOSS_CE, tanh=True, user_factor=False),
    "ce_tied_ada_gate": dict(loss=ob.LOSS_CE, linear_function=True),
    "sq_asym_ada_gate": dict(loss=ob.LOSS_SQUARE, asymmetric=True, linear_function=True),
}


def _flags(kw):
    kw = dict(kw)
    loss = kw.pop("loss")
    lr = kw.pop("learn_rate", 0.1)
    return loss, lr, kw


@pytest.mark.parametrize("name", list(VARIANTS))
def test_oracle_reproduces_step_kat(built, name):
    g = G("step_kat.npz")
    loss, lr, kw = _flags(VARIANTS[name])
    cfg = orc.OracleConfig(num_dim=int(g["K"]), loss_type=loss, learn_rate=lr, lambda_=0.01, corruption_ratio=0.5, beta=1.0, **kw)
    o = orc.Oracle(cfg, int(g["U"]), int(g["I"]), g["ptr"], g["col"])
    o.init_params(0)
    for k in range(12):
        if o.get(k).size:
            o.set(k, g[f"init_{k}"])
    z, y, gr, hg = o.step_user(int(g["uid"]), g["kept"], g["neg"])
    for got, key in ((z, "z"), (y, "y"), (gr, "g"), (hg, "hg")):
        np.testing.assert_allclose(got, g[f"{name}_{key}"], rtol=1e-13, atol=1e-15)
    for k in range(12):
        if o.get(k).size:
            np.testing.assert_allclose(o.get(k), g[f"{name}_after_{k}"].ravel(), rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("name", ["sq", "ce"])
def test_oracle_reproduces_loss_curve(built, name):
    g = G("loss_curve.npz")
    U, I = g["train_ptr"].size - 1, 100
    cfg = orc.OracleConfig(num_dim=int(g["K"]), loss_type=ob.LOSS_SQUARE if name == "sq" else ob.LOSS_CE, beta=1.0)
    o = orc.Oracle(cfg, U, I, g["train_ptr"], g["train_col"])
    o.init_params(0)
    for k in range(10):
        if f"{name}_init_{k}" in g:
            o.set(k, g[f"{name}_init_{k}"])
    seed = int(g["seed"])
    losses = []
    for ep in range(5):
        o.train_literal(seed, ep)
        losses.append(o.data_loss(seed, ep) + o.penalty_loss())
    np.testing.assert_allclose(losses, g[f"{name}_loss"], rtol=1e-12)
    np.testing.assert_array_equal(o.recommend(10), g[f"{name}_rec"])
    np.testing.assert_allclose(orc.eval_topn(g[f"{name}_rec"], g["test_ptr"], g["test_col"]), g[f"{name}_metrics"], rtol=1e-12)


def test_oracle_reproduces_loss_kat(built):
    g = G("loss_kat.npz")
    from cdae_amd import synth
    d = synth.generate_shape("tiny", seed=5)
    for name, lt in (("sq", ob.LOSS_SQUARE), ("ce", ob.LOSS_CE)):
        o = orc.Oracle(orc.OracleConfig(loss_type=lt), d.num_users, d.num_items, d.train_ptr, d.train_col)
        for t in (0, 1):
            np.testing.assert_allclose([o.loss_eval(p, float(t)) for p in g["pred"]], g[f"{name}_eval_t{t}"], rtol=1e-14)
            np.testing.assert_allclose([o.loss_grad(p, float(t)) for p in g["pred"]], g[f"{name}_grad_t{t}"], rtol=1e-14)


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VARIANTS))
def test_hip_reproduces_step_kat(built, name):
    """One explicit-input train_one_user_corruption (fixed mask, negatives with a duplicate) vs the golden
    post-step parameters.  fp32 path vs fp64 golden: |diff| <= 2e-6 + 3e-6 |ref| (a few fp32 ulps)."""
    g = G("step_kat.npz")
    loss, lr, kw = _flags(VARIANTS[name])
    K, U, I = int(g["K"]), int(g["U"]), int(g["I"])
    cfg = cdae_amd.CDAEConfig(num_dim=K, lt=loss, learn_rate=lr, lambda_=0.01, corruption_ratio=0.5, beta=1.0,
                              batch_users=1, **kw)
    m = cdae_amd.CDAE(cfg)
    m.set_interactions(U, I, g["ptr"], g["col"])
    present = [k for k in range(12) if f"{name}_after_{k}" in g]
    for k in present:
        m.set(k, g[f"init_{k}"])
    z = m.get_hidden_values([int(g["uid"])], mode=0)     # sanity: full-row encode is finite
    assert np.isfinite(z).all()
    m.train_one_user_corruption(int(g["uid"]), g["kept"], g["neg"])
    for k in present:
        np.testing.assert_allclose(m.get(k).ravel(), g[f"{name}_after_{k}"].ravel(), rtol=3e-6, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sq", "ce"])
def test_hip_reproduces_loss_curve(built, name):
    """Five epochs at batch_users = 1 (the reference's sequential schedule) on the committed 60 x 100
    dataset: loss per epoch within 1e-4 relative, final parameters within 5e-4 of their range, Recall@10
    and MAP@10 within 0.02 (a near-tie in a top-10 list may swap one id on 60 users)."""
    g = G("loss_curve.npz")
    U, I, K = g["train_ptr"].size - 1, 100, int(g["K"])
    cfg = cdae_amd.CDAEConfig(num_dim=K, lt=ob.LOSS_SQUARE if name == "sq" else ob.LOSS_CE, beta=1.0, batch_users=1)
    m = cdae_amd.CDAE(cfg)
    m.set_interactions(U, I, g["train_ptr"], g["train_col"])
    for k in range(10):
        if f"{name}_init_{k}" in g:
            m.set(k, g[f"{name}_init_{k}"])
    seed = int(g["seed"])
    losses = []
    for ep in range(5):
        m.train_one_iteration(seed, ep)
        losses.append(m.current_loss(seed, ep))
    np.testing.assert_allclose(losses, g[f"{name}_loss"], rtol=1e-4)
    for k in range(10):
        if f"{name}_final_{k}" in g:
            ref = g[f"{name}_final_{k}"].ravel()
            assert np.abs(m.get(k).ravel() - ref).max() < 5e-4 * (1e-3 + np.abs(ref).max()), k
    rec = m.recommend_all(10)
    gap = np.abs(np.diff(g[f"{name}_rec_scores"], axis=1)).min(axis=1)
    clear = gap > 1e-3
    np.testing.assert_array_equal(rec[clear], g[f"{name}_rec"][clear])
    met = orc.eval_topn(rec, g["test_ptr"], g["test_col"])
    assert np.abs(met - g[f"{name}_metrics"]).max() < 0.02


# ---- reference-sequenced mode: the reference's own generators in the reference's own order (tests/golden/make_ref_sequenced.py) ----
REF_PARAMS = (("W", ob.P_W, cdae_amd.P_W), ("W_ag", ob.P_W_AG, cdae_amd.P_W_AG), ("Wu", ob.P_WU, cdae_amd.P_WU), ("Wu_ag", ob.P_WU_AG, cdae_amd.P_WU_AG),
              ("b", ob.P_B, cdae_amd.P_B), ("b_ag", ob.P_B_AG, cdae_amd.P_B_AG), ("bp", ob.P_BP, cdae_amd.P_BP), ("bp_ag", ob.P_BP_AG, cdae_amd.P_BP_AG))


def _ref_cfg(g):
    c = {k: v for k, v in g["cfg"]}
    return dict(num_dim=int(c["num_dim"]), loss_type=int(c["loss_type"]), beta=float(c["beta"]), corruption_ratio=float(c["corruption_ratio"]),
                num_neg=int(c["num_neg"]), scaled=c["scaled"] == "True", learn_rate=float(c["learn_rate"]), lambda_=float(c["lambda_"]))


def test_oracle_reproduces_the_reference_sequenced_fixture(built):
    """srand(1) / mt19937_64 / unordered_map order -> the same initial values, the same draws for every user and the same
    parameters after one epoch as the committed file (regression pin of oracle_ref_*: recsys_model_base.hpp:46-57,
    random.hpp:34-37, cdae.hpp:112-120, 142, 217-220, 361-371)."""
    g = np.load(os.path.join(HERE, "golden", "ref_sequenced_tiny.npz"), allow_pickle=True)
    cfg = _ref_cfg(g)
    mk = lambda: orc.Oracle(orc.OracleConfig(**cfg), int(g["num_users"]), int(g["num_items"]), g["train_ptr"], g["train_col"])  # noqa: E731
    a, b = mk(), mk()
    for o in (a, b):
        o.ref_seed(int(g["mt_seed"]), int(g["rand_seed"]))
        o.ref_init_params()
    for name, w, _ in REF_PARAMS:
        assert np.array_equal(a.get(w), g[f"init_{name}"].ravel())
    pos, inp, neg = [], [], []
    for u in range(int(g["num_users"])):
        p, i, n = b.ref_draw_user(u)
        pos.append(p); inp.append(i); neg.append(n)
    assert np.array_equal(np.concatenate(pos), g["pos_order"])
    assert np.array_equal(np.concatenate(inp), g["inputs"]) and np.array_equal(np.concatenate(neg), g["negatives"])
    a.train_reference_sequenced()
    for name, w, _ in REF_PARAMS:
        np.testing.assert_allclose(a.get(w), g[f"final_{name}"].ravel(), rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_hip_trains_on_the_references_own_draws(built):
    """The HIP path fed the REFERENCE's draw sequence: every user's dropout mask and negatives as the reference's generators hand
    them out (committed fixture: glibc rand() after srand(1), mt19937_64, the unordered_map's visiting order), one user at a time
    through cdae_hip_train_one_user_corruption — the public per-user step of cdae.hpp:198-200 — starting from the rand()-drawn initial
    values.  After the epoch the fp32 device parameters are the fixture's fp64 ones to 2e-4 of each parameter's range.  No oracle
    code runs in this test."""
    g = np.load(os.path.join(HERE, "golden", "ref_sequenced_tiny.npz"), allow_pickle=True)
    cfg = _ref_cfg(g)
    U, I = int(g["num_users"]), int(g["num_items"])
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=cfg["num_dim"], lt=cdae_amd.CROSS_ENTROPY, beta=cfg["beta"], corruption_ratio=cfg["corruption_ratio"],
                                          num_neg=cfg["num_neg"], scaled=cfg["scaled"], learn_rate=cfg["learn_rate"], lambda_=cfg["lambda_"],
                                          batch_users=1))
    m.set_interactions(U, I, g["train_ptr"], g["train_col"])
    m.init_params(1)
    for name, _, w in REF_PARAMS:
        m.set(w, g[f"init_{name}"])
    tp, ip, nn = g["train_ptr"], g["in_ptr"], cfg["num_neg"]
    for u in range(U):
        negs = g["negatives"][tp[u] * nn:tp[u + 1] * nn]
        m.train_one_user_corruption(u, g["inputs"][ip[u]:ip[u + 1]], negs)
    for name, _, w in REF_PARAMS:
        ref = g[f"final_{name}"].ravel()
        err = np.abs(m.get(w).astype(np.float64).ravel() - ref).max() / (1e-3 + np.abs(ref).max())
        assert err < 2e-4, (name, err)
    m.close()
