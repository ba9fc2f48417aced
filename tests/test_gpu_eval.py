"""-m gpu: the TOPN metrics on the device (cdae_hip_set_test_rows / cdae_hip_eval_topn, since ABI 9) against the oracle's restatement of
TOPN_Evaluation::evaluate + evaluate_rec_list (/root/reference/src/model/evaluation.hpp:113-181, 183-219).

The lists are integer work and the eight columns are sums of per-user fp64 terms added in user order, so the bar is BIT equality:
  * the top-10 table the metric kernel scored is cdae_hip_recommend_all's table;
  * rets[8] == oracle.eval_topn(that table) as fp64 bit patterns (same expressions, same order of additions);
  * hits[3] == the integer hit counts numpy derives from the table.
Covered: the matrix-core recommend path over several 32 768-user chunks (ML-10M shape), the general per-user path (K > 256),
users without test items, IMF / BPR handles (one user per block, and the block schedule whose device rows are permuted), the
sharded handle (item-rows and user layouts), and the unmodified yelp app's table with the host loop against the device path.
"""
import os
import re
import subprocess

import numpy as np
import pytest

import cdae_amd
from cdae_amd import synth
import oracle as orc

pytestmark = pytest.mark.gpu


def numpy_hits(ids, test_ptr, test_col):
    h = np.zeros(3, dtype=np.uint64)
    for u in range(ids.shape[0]):
        row = test_col[test_ptr[u]:test_ptr[u + 1]]
        if row.size == 0:
            continue
        m = np.isin(ids[u], row)
        h += np.array([m[:1].sum(), m[:5].sum(), m[:10].sum()], dtype=np.uint64)
    return h


def check(m, d, topk=10):
    m.set_test_rows(d.test_ptr, d.test_col)
    rets, hits, ids = m.eval_topn(topk, with_ids=True)
    assert np.array_equal(ids, m.recommend_all(topk))
    ref = orc.eval_topn(ids, d.test_ptr, d.test_col)
    assert np.array_equal(rets.view(np.uint64), ref.view(np.uint64)), (rets, ref)
    rets2, hits2 = m.eval_topn(topk)                       # without the table: nothing but 16 numbers comes back
    assert np.array_equal(rets2.view(np.uint64), ref.view(np.uint64)) and np.array_equal(hits, hits2)
    return rets, hits, ids


@pytest.mark.parametrize("shape,K,B", [("tiny", 24, 32), ("small", 50, 64), ("small", 300, 64)])
def test_device_topn_metrics_are_the_oracles_bits(built, shape, K, B):
    d = synth.generate_shape(shape, seed=5)
    # a few users without test items (evaluation.hpp:139-140 skips them; n_test_users counts the others)
    tp = d.test_ptr.copy()
    drop = np.array([0, 7, d.num_users - 1])
    keep = np.ones(d.test_col.size, dtype=bool)
    for u in drop:
        keep[tp[u]:tp[u + 1]] = False
    lens = np.diff(d.test_ptr)
    lens[drop] = 0
    d2 = synth.Interactions(d.num_users, d.num_items, d.train_ptr, d.train_col, np.r_[0, np.cumsum(lens)].astype(np.int64), d.test_col[keep])
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, beta=1.0, batch_users=B))
    m.reset(d2, seed=3)
    for ep in range(2):
        m.train_one_iteration(3, ep)
        rets, hits, ids = check(m, d2)
        assert np.array_equal(hits, numpy_hits(ids, d2.test_ptr, d2.test_col))
    assert rets[5] > 0.05                                  # a trained model: Recall@10 is not degenerate
    # topk other than 10: P@10 / R@10 stay 0 below 10 places (evaluation.hpp:197-206), 20 places are scored at most (:186)
    for topk in (5, 16):
        check(m, d2, topk)
    m.close()


def test_device_topn_at_ml10m_shape_over_several_chunks(built):
    """70 000 users = three chunks of the matrix-core recommend path; the metric kernels run chunk by chunk on the same stream"""
    d = synth.generate_shape("ml10m", seed=20141119)
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, beta=1.0, batch_users=0))
    m.reset(d, seed=1)
    m.train_one_iteration(1, 0)
    rets, hits, ids = check(m, d)
    assert np.array_equal(hits, numpy_hits(ids, d.test_ptr, d.test_col))
    m.close()


@pytest.mark.parametrize("pairwise,B", [(False, 1), (True, 1), (False, 64)])
def test_device_topn_for_the_sibling_models(built, pairwise, B):
    d = synth.generate_shape("tiny", seed=5)
    m = cdae_amd.MF(cdae_amd.MFConfig(num_dim=16, lt=cdae_amd.SQUARE if not pairwise else cdae_amd.LOG, pairwise=pairwise, batch_users=B,
                                      learn_rate=0.1, beta=1.0, lambda_=0.01, num_neg=5))
    m.reset(d, seed=11)
    m.train_one_iteration(11, 0)
    check(m, d)
    m.close()


@pytest.mark.parametrize("item_rows", [True, False])
def test_sharded_handle_sums_the_same_columns(built, item_rows):
    d = synth.generate_shape("small", seed=5)
    cfg = cdae_amd.CDAEConfig(num_dim=50, lt=cdae_amd.CROSS_ENTROPY, beta=1.0, batch_users=64)
    m = cdae_amd.MultiCDAE(cfg, devices=[0, 0, 0], item_rows=item_rows)
    m.reset(d, seed=3)
    m.train_one_iteration(3, 0)
    rets, hits = m.eval_topn(d.test_ptr, d.test_col)
    ids = m.recommend_all(10)
    ref = orc.eval_topn(ids, d.test_ptr, d.test_col)
    assert np.array_equal(rets.view(np.uint64), ref.view(np.uint64))
    assert np.array_equal(hits, numpy_hits(ids, d.test_ptr, d.test_col))
    m.close()


def test_set_test_rows_rejects_bad_rows(built):
    d = synth.generate_shape("tiny", seed=5)
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=8, lt=cdae_amd.SQUARE, batch_users=16))
    with pytest.raises(cdae_amd.CDAEError):
        m.set_test_rows(d.test_ptr, d.test_col)            # before set_interactions
    m.reset(d, seed=3)
    with pytest.raises(cdae_amd.CDAEError):
        m.eval_topn(10)                                    # before set_test_rows
    bad = d.test_col.copy()
    a, b = d.test_ptr[1], d.test_ptr[2]
    if b - a >= 2:
        bad[a], bad[a + 1] = bad[a + 1], bad[a]
        with pytest.raises(cdae_amd.CDAEError):
            m.set_test_rows(d.test_ptr, bad)               # a row that is not ascending
    bad = d.test_col.copy()
    bad[0] = d.num_items
    with pytest.raises(cdae_amd.CDAEError):
        m.set_test_rows(d.test_ptr, bad)                   # item id out of range
    m.close()


def test_reference_yelp_app_prints_the_same_topn_columns_from_the_device(built, tmp_path):
    """TOPN_Evaluation<CDAE>::evaluate of the host layer takes the device path when the model was reset with `train`
    (src/model/evaluation.hpp); CDAE_HOST_TOPN=1 keeps the host loop.  Same seed -> the printed columns are identical strings."""
    from tests.test_host_cpp import write_ratings, run
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    yelp = os.path.join(root, "build", "yelp")
    if not os.path.exists(yelp):
        pytest.skip("no build/yelp (reference sources were not present at build time)")
    write_ratings(tmp_path / "yelp_10core.txt")
    for task in ("prepare", "split"):
        assert run([yelp, f"--task={task}"], tmp_path)[0] == 255
    tables = []
    for env in ({}, {"CDAE_HOST_TOPN": "1"}):
        rc, out = run([yelp, "--task=test", "--method=CDAE", "--num_dim=50", "--loss_type=CE", "--cratio=0.5", "--scaled=true", "--beta=1"],
                      tmp_path, env={"CDAE_SEED": "11", "CDAE_BATCH_USERS": "64", **env})
        assert rc == 0, out[-3000:]
        rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)]
        assert len(rows) == 2 + 51
        tables.append([r.split("|")[2:10] for r in rows[2:]])          # Train Loss + the eight TOPN columns (not the time columns)
    assert tables[0] == tables[1]
