"""GPU parity on degenerate shapes and flag combinations (through the C ABI, against the oracle's batched schedule):
fewer items than a 32-item tile or the 8 gather partitions, one-item users, batch larger than the data set, K from 1 to 512,
no negatives, nothing dropped, several corruptions."""
import numpy as np
import pytest

import cdae_amd
from cdae_amd import synth
from helpers import make_pair, max_param_err

pytestmark = pytest.mark.gpu

SHAPES = [(3, 5, 1, 2), (40, 9, 1, 4), (130, 33, 1, 10), (70, 300, 1, 40)]
VARIANTS = [dict(K=1, B=1), dict(K=3, B=7), dict(K=64, B=1000), dict(K=200, B=32, num_neg=0), dict(K=17, B=16, num_neg=1),
            dict(K=40, B=8, corruption_ratio=0.0), dict(K=512, B=4, user_factor=False),
            dict(K=33, B=5, loss=cdae_amd.SQUARE, learn_rate=0.02), dict(K=128, B=64, using_adagrad=False, learn_rate=0.01),
            dict(K=256, B=9, asymmetric=True), dict(K=65, B=3, tanh=True), dict(K=20, B=6, linear=True, learn_rate=0.01),
            dict(K=24, B=4, num_corruptions=2)]


def _data(U, I, lo, hi, seed):
    rng = np.random.default_rng(seed)
    rows = [np.sort(rng.choice(I, size=int(rng.integers(lo, hi + 1)), replace=False)).astype(np.uint32) for _ in range(U)]
    ptr = np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.int64)
    return synth.Interactions(U, I, ptr, np.concatenate(rows), np.zeros(U + 1, dtype=np.int64), np.zeros(0, dtype=np.uint32))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d" % s[:2])
@pytest.mark.parametrize("variant", VARIANTS, ids=lambda v: "-".join(f"{k}{v[k]}" for k in sorted(v)))
def test_degenerate_shapes_track_the_oracle(built, shape, variant):
    d = _data(*shape, seed=17)
    model, o = make_pair(d, **variant)
    B = variant["B"]
    for ep in range(2):
        model.train_one_iteration(seed=4, epoch=ep)
        o.train_batched(4, ep, B)
    err, which = max_param_err(model, o)
    assert err < 5e-4, (err, which)
    lg, lo = model.data_loss(5, 0), o.data_loss(5, 0)
    assert abs(lg - lo) <= 5e-4 * max(1.0, abs(lo))
    topk = max(1, min(10, d.num_items - int(np.diff(d.train_ptr).max())))
    rec = model.recommend_all(topk)
    assert rec.shape == (d.num_users, topk)
    for u in range(d.num_users):                              # distinct, unrated, in range
        rated = d.train_col[d.train_ptr[u]:d.train_ptr[u + 1]]
        assert len(set(rec[u].tolist())) == topk and rec[u].max() < d.num_items and not np.intersect1d(rec[u], rated).size
    with pytest.raises(cdae_amd.CDAEError):
        cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=8, corruption_ratio=1.0, scaled=True)).reset(d, seed=1)
