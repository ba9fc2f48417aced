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


def make(d, *, K=16, B=1, loss=cdae_amd.SQUARE, pairwise=False, seed=11, **kw):
    hyper = dict(learn_rate=0.1, beta=1.0, lambda_=0.01, num_neg=5, using_bias_term=True, using_adagrad=True)
    hyper.update(kw)
    m = cdae_amd.MF(cdae_amd.MFConfig(num_dim=K, lt=loss, pairwise=pairwise, batch_users=B, **hyper))
    m.reset(d, seed=seed)
    o = orc.MfOracle(orc.MfConfig(num_dim=K, loss_type=loss, pairwise=pairwise, **hyper), d.num_users, d.num_items, d.train_ptr, d.train_col)
    o.init_params(seed)
    for po, pg in PAIRS:                                   # start both from the device's fp32 parameters
        np.testing.assert_allclose(m.get(pg).astype(np.float64).ravel(), o.get(po), rtol=1e-6, atol=1e-9)   # same init stream
        o.set(po, m.get(pg).astype(np.float64))
    return m, o


def max_err(m, o):
    worst, name = 0.0, None
    for po, pg in PAIRS:
        ref = o.get(po)
        err = np.abs(m.get(pg).astype(np.float64).ravel() - ref).max() / (1e-3 + np.abs(ref).max())
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
    rec_g = m.recommend_all(10)
    rec_o, sc = o.recommend(10, with_scores=True)
    clear = np.abs(np.diff(sc, axis=1)).min(axis=1) > 1e-4
    assert clear.mean() > 0.9
    np.testing.assert_array_equal(rec_g[clear], rec_o[clear])
    rec20 = m.recommend_all(20)                               # topk > 16: the general recommend path
    np.testing.assert_array_equal(rec20[clear][:, :10], rec_o[clear])
    r_g = orc.eval_topn(rec_g, d.test_ptr, d.test_col)[5]
    r_o = orc.eval_topn(rec_o, d.test_ptr, d.test_col)[5]
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
