"""-m gpu: the sibling SGD models IMF and BPR on the GPU, through the C ABI (cdae_hip_create_mf),
against oracle/mf_oracle.cpp — the fp64 restatement of /root/reference/src/model/recsys/imf.hpp:57-119 and bpr.hpp:56-106.

  * batch_users = 1 is the reference loop (users in order, every instance steps the user vector and the item row at once,
    a user's duplicate negatives included): compared with the oracle's LITERAL schedule;
  * batch_users > 1 is the block schedule (phase U per user against the block-start item rows, phase I per item row in (user,
    instance) order): compared with the oracle's same schedule;
  * every loss yelp.cpp:122-165 lets the two models choose, AdaGrad and plain SGD, with and without bias terms.
Tolerance: fp32 storage and 1-ulp hardware rcp / sqrt / exp against fp64 — 2e-4 of the parameter's range after two epochs,
like the CDAE parity tests.
"""
import glob
import os

import numpy as np
import pytest

import cdae_amd
from cdae_amd import synth
import oracle as orc
from oracle import binding as ob

pytestmark = pytest.mark.gpu

PAIRS = [(ob.MF_UV, cdae_amd.P_WU), (ob.MF_UV_AG, cdae_amd.P_WU_AG), (ob.MF_IV, cdae_amd.P_W), (ob.MF_IV_AG, cdae_amd.P_W_AG),
         (ob.MF_UB, cdae_amd.P_UB), (ob.MF_UB_AG, cdae_amd.P_UB_AG), (ob.MF_IB, cdae_amd.P_BP), (ob.MF_IB_AG, cdae_amd.P_BP_AG)]


@pytest.fixture(scope="module")
def tiny(built):
    return synth.generate_shape("tiny", seed=5)


USER_INDEXED = {cdae_amd.P_WU, cdae_amd.P_WU_AG, cdae_amd.P_UB, cdae_amd.P_UB_AG}


def by_position(m, which):
    """a parameter of the device model in the order the ORACLE holds it: user-indexed arrays in the handle's training order"""
    a = m.get(which).astype(np.float64)
    return a[m.user_order()] if which in USER_INDEXED else a


def in_training_order(d, order):
    """the data set with its users renumbered by training position (what the oracle trains on: its user t = the handle's t-th user)"""
    def cut(ptr, col):
        lens = np.diff(ptr)[order]
        p = np.r_[0, np.cumsum(lens)].astype(np.int64)
        c = np.concatenate([col[ptr[u]:ptr[u + 1]] for u in order]) if order.size else col
        return p, c.astype(np.uint32)
    tp, tc = cut(d.train_ptr, d.train_col)
    sp, sc = cut(d.test_ptr, d.test_col)
    return synth.Interactions(d.num_users, d.num_items, tp, tc, sp, sc)


def make(d, *, K=16, B=1, loss=cdae_amd.SQUARE, pairwise=False, seed=11, **kw):
    hyper = dict(learn_rate=0.1, beta=1.0, lambda_=0.01, num_neg=5, using_bias_term=True, using_adagrad=True)
    hyper.update(kw)
    m = cdae_amd.MF(cdae_amd.MFConfig(num_dim=K, lt=loss, pairwise=pairwise, batch_users=B, **hyper))
    m.reset(d, seed=seed)
    # the block schedule (B > 1) trains in activity-grouped order (cdae_hip_user_order): the oracle gets the users in THAT order
    order = m.user_order()
    assert sorted(order.tolist()) == list(range(d.num_users))
    if B == 1 or d.num_users <= B:
        np.testing.assert_array_equal(order, np.arange(d.num_users))      # the reference's order
    dp = in_training_order(d, order)
    o = orc.MfOracle(orc.MfConfig(num_dim=K, loss_type=loss, pairwise=pairwise, **hyper), dp.num_users, dp.num_items, dp.train_ptr, dp.train_col)
    o.init_params(seed)
    for po, pg in PAIRS:                                   # start both from the device's fp32 parameters
        np.testing.assert_allclose(by_position(m, pg).ravel(), o.get(po), rtol=1e-6, atol=1e-9)   # same init stream, keyed by position
        o.set(po, by_position(m, pg))
    return m, o


def max_err(m, o):
    worst, name = 0.0, None
    for po, pg in PAIRS:
        ref = o.get(po)
        err = np.abs(by_position(m, pg).ravel() - ref).max() / (1e-3 + np.abs(ref).max())
        if err > worst:
            worst, name = err, pg
    return worst, name


IMF_VARIANTS = [dict(), dict(loss=cdae_amd.CROSS_ENTROPY), dict(loss=cdae_amd.LOG), dict(loss=cdae_amd.HINGE),
                dict(using_adagrad=False, learn_rate=0.02), dict(using_bias_term=False), dict(num_neg=1), dict(beta=0.5, lambda_=0.05)]
BPR_VARIANTS = [dict(loss=cdae_amd.LOG), dict(loss=cdae_amd.SQUARE), dict(loss=cdae_amd.HINGE), dict(loss=cdae_amd.LOG, using_adagrad=False, learn_rate=0.02),
                dict(loss=cdae_amd.LOG, using_bias_term=False), dict(loss=cdae_amd.LOG, num_neg=2)]


@pytest.mark.parametrize("variant", IMF_VARIANTS, ids=lambda v: "-".join(f"{k}{v[k]}" for k in sorted(v)) or "default")
def test_imf_sequential_schedule_is_the_reference_loop(tiny, variant):
    m, o = make(tiny, K=24, B=1, **variant)
    for ep in range(2):
        m.train_one_iteration(seed=7, epoch=ep)
        o.train_literal(7, ep)
    err, which = max_err(m, o)
    assert err < 2e-4, (err, which)


@pytest.mark.parametrize("variant", BPR_VARIANTS, ids=lambda v: "-".join(f"{k}{v[k]}" for k in sorted(v)))
def test_bpr_sequential_schedule_is_the_reference_loop(tiny, variant):
    m, o = make(tiny, K=24, B=1, pairwise=True, **variant)
    for ep in range(2):
        m.train_one_iteration(seed=7, epoch=ep)
        o.train_literal(7, ep)
    err, which = max_err(m, o)
    assert err < 2e-4, (err, which)


@pytest.mark.parametrize("pairwise,loss", [(False, cdae_amd.SQUARE), (False, cdae_amd.CROSS_ENTROPY), (True, cdae_amd.LOG), (True, cdae_amd.HINGE)])
@pytest.mark.parametrize("B,K", [(7, 10), (64, 64), (300, 65), (50, 200), (33, 300)])
def test_block_schedule_matches_oracle(tiny, pairwise, loss, B, K):
    m, o = make(tiny, K=K, B=B, loss=loss, pairwise=pairwise)
    for ep in range(2):
        st = m.train_one_iteration(seed=3, epoch=ep)
        o.train_batched(3, ep, B)
        assert st.users == tiny.num_users
    err, which = max_err(m, o)
    assert err < 2e-4, (err, which)


@pytest.mark.parametrize("pairwise", [False, True])
def test_recommend_and_reported_loss(built, pairwise):
    d = synth.generate(1200, 500, 60_000, seed=9)
    m, o = make(d, K=32, B=64, loss=cdae_amd.LOG if pairwise else cdae_amd.SQUARE, pairwise=pairwise)
    for ep in range(3):
        m.train_one_iteration(seed=2, epoch=ep)
        o.train_batched(2, ep, 64)
    order = m.user_order()
    assert not np.array_equal(order, np.arange(d.num_users))  # 1200 users in blocks of 64: activity-grouped
    rec_g = m.recommend_all(10)                               # by user id
    rec_o, sc = o.recommend(10, with_scores=True)             # by training position
    clear = np.abs(np.diff(sc, axis=1)).min(axis=1) > 1e-4
    assert clear.mean() > 0.9
    np.testing.assert_array_equal(rec_g[order][clear], rec_o[clear])
    rec20 = m.recommend_all(20)                               # topk > 16: the general recommend path
    np.testing.assert_array_equal(rec20[order][clear][:, :10], rec_o[clear])
    np.testing.assert_array_equal(m.recommend_all(10, 100, 140), rec_g[100:140])     # a range of user ids
    dp = in_training_order(d, order)
    r_g = orc.eval_topn(rec_g, d.test_ptr, d.test_col)[5]
    r_o = orc.eval_topn(rec_o, dp.test_ptr, dp.test_col)[5]
    assert abs(r_g - r_o) < 2e-3 and r_g > 0.1                # the model learned something
    assert m.current_loss(1, 0) == 0.0                        # ModelBase::data_loss / penalty_loss defaults (model_base.hpp:36-45)


def test_user_with_duplicate_negatives_and_many_items(built):
    """few items: a user draws the same negative several times — in the sequential schedule the second visit must see the first's step"""
    rng = np.random.default_rng(2)
    rows = [np.sort(rng.choice(12, n, replace=False)).astype(np.uint32) for n in (3, 8, 5, 9, 2, 7)]
    ptr = np.r_[0, np.cumsum([r.size for r in rows])].astype(np.int64)
    d = synth.Interactions(len(rows), 12, ptr, np.concatenate(rows), np.zeros(len(rows) + 1, np.int64), np.empty(0, np.uint32))
    for pairwise in (False, True):
        m, o = make(d, K=8, B=1, pairwise=pairwise, loss=cdae_amd.LOG if pairwise else cdae_amd.SQUARE)
        for ep in range(3):
            m.train_one_iteration(seed=5, epoch=ep)
            o.train_literal(5, ep)
        err, which = max_err(m, o)
        assert err < 2e-4, (pairwise, err, which)


def test_entry_points_that_do_not_apply_fail_loudly(tiny):
    m, _ = make(tiny, K=8, B=16)
    with pytest.raises(cdae_amd.CDAEError):
        m.get_hidden_values([0, 1])
    with pytest.raises(cdae_amd.CDAEError):
        m.train_one_user_corruption(0, [], [])
    with pytest.raises(cdae_amd.CDAEError):
        cdae_amd.MF(cdae_amd.MFConfig(lt=4))                 # SQUARED_HINGE: not a loss yelp.cpp offers these models


def test_block_schedule_orders_users_by_activity_and_round_trips_parameters(built):
    """batch_users > 1: blocks of users with similar train-row lengths (a block lasts as long as its most active user's chain), the
    blocks in a fixed pseudo-random order; get / set_param stay by user id."""
    d = synth.generate(1280, 500, 64_000, seed=9)
    m = cdae_amd.MF(cdae_amd.MFConfig(num_dim=16, batch_users=64))
    m.reset(d, seed=3)
    order = m.user_order()
    n = np.diff(d.train_ptr)[order].reshape(-1, 64)            # 20 full blocks
    assert (n.max(axis=1) - n.min(axis=1)).max() <= np.diff(d.train_ptr).max() // 4            # similar lengths inside a block
    firsts = n[:, 0]
    assert not (np.diff(firsts) <= 0).all() and not (np.diff(firsts) >= 0).all()                # heavy and light blocks interleave
    rng = np.random.default_rng(0)
    wu = rng.standard_normal((d.num_users, 16)).astype(np.float32)
    ub = rng.standard_normal(d.num_users).astype(np.float32)
    m.set(cdae_amd.P_WU, wu); m.set(cdae_amd.P_UB, ub)
    np.testing.assert_array_equal(m.get(cdae_amd.P_WU), wu)
    np.testing.assert_array_equal(m.get(cdae_amd.P_UB), ub)
    one = cdae_amd.MF(cdae_amd.MFConfig(num_dim=16, batch_users=1))
    one.reset(d, seed=3)
    np.testing.assert_array_equal(one.user_order(), np.arange(d.num_users))


@pytest.mark.parametrize("pairwise,loss", [(False, cdae_amd.SQUARE), (True, cdae_amd.LOG)])
def test_block_schedule_recall_against_the_sequential_loop(built, pairwise, loss):
    """Accuracy cost of LARGE blocks (256 users: an explicit throughput setting; the defaults are 16 users per block for IMF — certified
    below — and one for BPR): Recall@10 after
    four epochs against the SEQUENTIAL loop (batch_users = 1 = the reference's, imf.hpp:71-115 / bpr.hpp:56-106) on the same data,
    averaged over three stream seeds."""
    d = synth.generate_shape("small", seed=21)
    out = {}
    for B in (1, 256):
        recs = []
        for seed in (1, 2, 3):
            m = cdae_amd.MF(cdae_amd.MFConfig(num_dim=32, lt=loss, pairwise=pairwise, batch_users=B))
            m.reset(d, seed=seed)
            for ep in range(4):
                m.train_one_iteration(seed, ep)
            recs.append(orc.eval_topn(m.recommend_all(10), d.test_ptr, d.test_col)[5])
            m.close()
        out[B] = np.array(recs)
    print(f"\n{'BPR' if pairwise else 'IMF'} recall@10 after 4 epochs: sequential {np.round(out[1], 4)} block of 256 {np.round(out[256], 4)}")
    assert out[1].mean() > 0.12                                # both learn (Popularity: ~0.09 on this shape)
    # IMF: 256-user blocks within 0.01 of the loop (measured 0.000 ... +0.005).  BPR: the block schedule is AHEAD of the loop at this size
    # (0.240 against 0.226 after four epochs on this 4 000-user data set: summed steps, as with small full-output blocks);
    # bounded on both sides
    assert -0.01 <= out[256].mean() - out[1].mean() <= (0.02 if pairwise else 0.01)


# ---- the library defaults of the sibling models, held to the accuracy bound of the sampled CDAE path ------------------------------
# Anchors: `ml10m_k200_{imf,bpr}_seq_seed*.npz` — Recall@10 of the SEQUENTIAL loop (imf.hpp:71-115 / bpr.hpp:56-106; the fp64 oracle's
# literal restatement, make_mf_literal_curves.py: ~3 min of one core per epoch with its evaluation) at ML-10M shape K=200, six seeds x
# five epochs.
_GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
IMF_SEQ = sorted(glob.glob(os.path.join(_GOLD, "ml10m_k200_imf_seq_seed*.npz")))
BPR_SEQ = sorted(glob.glob(os.path.join(_GOLD, "ml10m_k200_bpr_seq_seed*.npz")))


_ML10M = {}


def _ml10m_data(seed):                   # (the two models' runs share the six synthetic data sets: ~5 s each to generate)
    if seed not in _ML10M:
        _ML10M[seed] = synth.generate_shape("ml10m", seed=seed)
    return _ML10M[seed]


@pytest.mark.parametrize("name", ["IMF", "BPR"])
def test_library_default_block_holds_the_accuracy_bound_at_ml10m_shape(built, name):
    """north_star's tolerance as the sampled CDAE path states it: Recall@10 within +-0.002 of the reference loop at every
    epoch AS A MEAN OVER SIX SEEDS (a single seed's difference is seed noise: bounded at 0.007).  The handle is created with
    batch_users = 0, i.e. this is whatever the library ships: 16 users per block for IMF, 8 for BPR at this size
    (cdae_hip_mf_default_batch_users; tools/mf_envelope.py has the other block sizes and Yelp shape — IMF at 32 users per block: mean offset
    up to +0.0035, 64: +0.005; BPR at Yelp shape's 10 000 users: +0.003 in the first two epochs, which is why its default is the loop
    below 65 536 users)."""
    pairwise = name == "BPR"
    fixtures = BPR_SEQ if pairwise else IMF_SEQ
    assert len(fixtures) >= 6
    diffs = []
    for p in fixtures:
        f = np.load(p, allow_pickle=True)
        seed = int(f["seed"])
        d = _ml10m_data(seed)
        assert d.nnz_train == int(f["nnz_train"]) and str(f["model"]) == name
        m = cdae_amd.MF(cdae_amd.MFConfig(num_dim=200, lt=cdae_amd.LOG if pairwise else cdae_amd.SQUARE, pairwise=pairwise, batch_users=0))
        m.reset(d, seed=seed)
        assert m.batch_users == (cdae_amd.binding.BPR_DEFAULT_BATCH_USERS if pairwise else cdae_amd.binding.IMF_DEFAULT_BATCH_USERS)
        m.set_test_rows(d.test_ptr, d.test_col)
        rec = []
        for ep in range(len(f["recall10"])):
            m.train_one_iteration(seed, ep)
            rec.append(m.eval_topn(10)[0][5])
        m.close()
        diffs.append(np.array(rec) - f["recall10"])
    diffs = np.array(diffs)
    print(f"\n{name}, ML-10M shape, {len(fixtures)} seeds, library default block: mean signed dRecall@10 per epoch {np.round(diffs.mean(axis=0), 5)}, "
          f"max |d| per epoch {np.round(np.abs(diffs).max(axis=0), 5)}")
    assert np.abs(diffs.mean(axis=0)).max() <= 0.002
    assert np.abs(diffs).max() <= 0.007


def test_small_data_sets_keep_the_reference_loop_by_default(built):
    d = synth.generate_shape("tiny", seed=5)
    for pairwise in (False, True):
        m = cdae_amd.MF(cdae_amd.MFConfig(num_dim=8, pairwise=pairwise, lt=cdae_amd.LOG if pairwise else cdae_amd.SQUARE, batch_users=0))
        m.reset(d, seed=3)
        assert m.batch_users == 1
        np.testing.assert_array_equal(m.user_order(), np.arange(d.num_users))
        m.close()
