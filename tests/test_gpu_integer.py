"""-m gpu: BIT-EXACT parity of the integer work of the hot path (tier bar: integer / index work is compared with
assert_array_equal, never through fp32 outcomes).

cdae_hip_debug_sample_batch copies back what one batch's sampling + sorting + segmentation produced on the device;
expected values come from the CPU oracle's draws (oracle_draw_inputs = get_corrputed_input cdae.hpp:361-371,
oracle_draw_negatives = sample_negative_item recsys_model_base.hpp:46-57 as called at cdae.hpp:217-220) and from
numpy's stable argsort for the item-major order (users in order inside an item = the order the row's chain of
updates runs in).
"""
import numpy as np
import pytest

import cdae_amd
from cdae_amd import synth
import oracle as orc
from oracle import binding as ob

pytestmark = pytest.mark.gpu

SLOT_MASK, DUP_PREV, DUP_NEXT, TARGET, INPUT = 0x0FFFFFFF, 1 << 28, 1 << 29, 1 << 30, 1 << 31
NONE = 0xFFFFFFFF


def expected_examples(o, d, seed, epoch, u0, nb, cidx, num_neg):
    """user-major (item, word) lists of the batch from the oracle's draws"""
    items, words = [], []
    for s in range(nb):
        u = u0 + s
        row = d.train_col[d.train_ptr[u]:d.train_ptr[u + 1]]
        kept = set(o.draw_inputs(seed, epoch, u, cidx).tolist())
        neg = o.draw_negatives(seed, epoch, u, cidx) if num_neg else np.empty(0, np.uint32)
        if num_neg and cidx:       # draw_negatives returns corruption cidx's draws
            pass
        items.append(row)
        words.append(np.array([s | TARGET | (INPUT if int(i) in kept else 0) for i in row], dtype=np.uint64))
        items.append(neg)
        words.append(np.full(neg.size, s, dtype=np.uint64))
    return np.concatenate(items).astype(np.uint32), np.concatenate(words)


def check_batch(model, o, d, seed, epoch, u0, nb, cidx=0, num_neg=5):
    got = model.debug_sample_batch(seed, epoch, u0, nb, cidx)
    items, words = expected_examples(o, d, seed, epoch, u0, nb, cidx, num_neg)
    E = items.size
    assert got["ex_item"].size == E
    # ---- masks and negatives: bit-exact against the oracle's draws
    np.testing.assert_array_equal(got["ex_item"], items)
    np.testing.assert_array_equal(got["ex_val"] >> np.uint64(32), np.arange(E, dtype=np.uint64))      # example index
    np.testing.assert_array_equal(got["ex_val"] & np.uint64(0xFFFFFFFF), words)
    # ---- item-major order: a STABLE sort by item of the user-major list
    order = np.argsort(items, kind="stable")
    np.testing.assert_array_equal(got["sorted_item"], items[order])
    np.testing.assert_array_equal(got["sorted_val"] >> np.uint64(32), order.astype(np.uint64))
    flags_mask = np.uint64(0xFFFFFFFF & ~(DUP_PREV | DUP_NEXT))
    np.testing.assert_array_equal(got["sorted_val"] & flags_mask, words[order])
    # ---- segments
    si = items[order]
    seg_b = np.zeros(d.num_items, np.uint32)
    seg_e = np.zeros(d.num_items, np.uint32)
    first = np.flatnonzero(np.r_[True, si[1:] != si[:-1]])
    last = np.flatnonzero(np.r_[si[1:] != si[:-1], True])
    seg_b[si[first]] = first
    seg_e[si[last]] = last + 1
    np.testing.assert_array_equal(got["seg_begin"], seg_b)
    np.testing.assert_array_equal(got["seg_end"], seg_e)
    # ---- duplicate negatives of one user inside a row: flags exact; numbering: distinct correction rows (the index space is
    # striped over DUP_STRIPES counters, cdae_kernels.hpp, so the numbers are not dense; a stripe that runs out hands out NONE)
    slot = (words[order] & np.uint64(SLOT_MASK)).astype(np.int64)
    same_prev = np.r_[False, (si[1:] == si[:-1]) & (slot[1:] == slot[:-1])]
    same_next = np.r_[same_prev[1:], False]
    w = got["sorted_val"].astype(np.uint64)
    np.testing.assert_array_equal((w & np.uint64(DUP_PREV)) != 0, same_prev)
    np.testing.assert_array_equal((w & np.uint64(DUP_NEXT)) != 0, same_next)
    n_dup = int(same_prev.sum())
    numbered = got["dup_of_pos"][same_prev]
    given = numbered[numbered != NONE]
    assert np.unique(given).size == given.size                      # no correction row is shared
    assert given.size >= min(n_dup, 8)                              # and the stripes did hand rows out
    assert (got["dup_of_pos"][~same_prev] == NONE).all()
    exp_of_ex = np.full(E, NONE, dtype=np.uint32)
    exp_of_ex[order[same_prev]] = numbered
    np.testing.assert_array_equal(got["dup_of_ex"], exp_of_ex)
    return n_dup


def make(d, K=8, B=64, seed=11, **kw):
    hyper = dict(num_neg=5, num_corruptions=1, corruption_ratio=0.5, scaled=True, learn_rate=0.1, beta=1.0, lambda_=0.01)
    hyper.update(kw)
    model = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, batch_users=B, **hyper))
    model.reset(d, seed=seed)
    o = orc.Oracle(orc.OracleConfig(num_dim=K, loss_type=ob.LOSS_CE, **hyper), d.num_users, d.num_items, d.train_ptr, d.train_col)
    return model, o


def test_masks_negatives_and_sort_bit_exact_small(built):
    d = synth.generate(1200, 500, 60_000, seed=9)
    model, o = make(d, B=96)
    dups = 0
    for ep, u0 in ((0, 0), (3, 96), (1, 1200 - 96)):
        dups += check_batch(model, o, d, 20141119, ep, u0, 96)
    assert dups > 0                     # 500 items: duplicate negatives do occur and are numbered
    check_batch(model, o, d, 5, 0, 7, 1)           # a one-user batch (the sequential schedule)


@pytest.mark.parametrize("q,num_neg", [(0.0, 1), (1.0, 2), (0.3, 7)])
def test_corruption_ratio_and_num_neg_variants(built, q, num_neg):
    d = synth.generate_shape("tiny", seed=5)
    model, o = make(d, B=50, corruption_ratio=q, scaled=q < 1.0 and q > 0.0, num_neg=num_neg)
    check_batch(model, o, d, 77, 2, 10, 50, num_neg=num_neg)


def test_second_corruption_draws_its_own_streams(built):
    d = synth.generate_shape("tiny", seed=5)
    hyper = dict(num_corruptions=3)
    model, o = make(d, B=40, **hyper)
    a = model.debug_sample_batch(9, 1, 0, 40, 0)
    b = model.debug_sample_batch(9, 1, 0, 40, 2)
    assert not np.array_equal(a["ex_item"], b["ex_item"])
    check_batch(model, o, d, 9, 1, 0, 40, cidx=2)


def test_user_with_more_than_2048_items_and_a_user_who_rated_almost_everything(built):
    """sample_kernel stages rows of <= 2048 items in LDS and searches longer ones in global memory; a user who rated all
    but 3 items exhausts the 32 rejection tries and takes the deterministic forward walk (include/cdae_rng.h)."""
    rng = np.random.default_rng(3)
    I = 6000
    rows = [np.sort(rng.choice(I, n, replace=False)).astype(np.uint32) for n in (2500, 30, I - 3, 2049, 17)]
    ptr = np.r_[0, np.cumsum([r.size for r in rows])].astype(np.int64)
    d = synth.Interactions(len(rows), I, ptr, np.concatenate(rows), np.zeros(len(rows) + 1, np.int64), np.empty(0, np.uint32))
    model, o = make(d, B=5, num_neg=2)
    check_batch(model, o, d, 1, 0, 0, 5, num_neg=2)
    check_batch(model, o, d, 1, 4, 2, 3, num_neg=2)


def test_more_than_65536_items_uses_32_bit_keys(built):
    d = synth.generate(300, 70_000, 12_000, seed=4, min_items=20)
    model, o = make(d, B=128)
    check_batch(model, o, d, 3, 0, 100, 128)


def test_ml10m_shape_batch_at_bench_batch_users(built):
    """one 512-user batch at the BASELINE shape (ex. 350 K examples; the oracle draws them in ~2 s)"""
    d = synth.generate_shape("ml10m", seed=20141119)
    model, o = make(d, K=8, B=512)
    check_batch(model, o, d, 20141119, 0, 512 * 57, 512)


def test_user_id_offset_shifts_the_streams_like_a_global_run(built):
    """a data-parallel shard draws what the single-GPU run draws for the same global users"""
    d = synth.generate_shape("tiny", seed=5)
    whole, o = make(d, B=60)
    lo, hi = 120, 180
    shard = d.user_range(lo, hi)
    m2 = cdae_amd.CDAE(whole.cfg)
    m2.set_interactions(shard.num_users, shard.num_items, shard.train_ptr, shard.train_col, user_id_offset=lo)
    a = whole.debug_sample_batch(4, 1, lo, 60)
    b = m2.debug_sample_batch(4, 1, 0, 60)
    for k in ("ex_item", "ex_val", "sorted_item", "sorted_val", "seg_begin", "seg_end"):
        np.testing.assert_array_equal(a[k], b[k])


@pytest.mark.parametrize("path", ["bucket", "scan", "tile", "library"])
def test_every_sort_path_gives_the_same_bit_exact_order(built, monkeypatch, devlib, path):
    """Default: bucket_sort_kernel (cdae_sort_kernels.hpp: one narrow launch, every workgroup owns a range of item ids and reads the
    cells sample_kernel routed its examples into).  CDAE_SORT_SCAN=1 (developer build): the same kernel without cells — every workgroup
    scans the batch's key list (what IMF / BPR handles and overflowing batches take).  CDAE_SORT_TILE=1
    (developer build): the per-tile LDS counting sort + per-item ordering, four launches.  CDAE_SORT_LIBRARY=1: the library radix sort +
    segment_kernel (what larger batches / item spaces still take).  All three must equal numpy's stable sort by item, bit for bit."""
    if path == "scan":
        monkeypatch.setenv("CDAE_SORT_SCAN", "1")
    if path == "tile":
        monkeypatch.setenv("CDAE_SORT_TILE", "1")
    if path == "library":
        monkeypatch.setenv("CDAE_SORT_LIBRARY", "1")
    d = synth.generate(1200, 500, 60_000, seed=9)
    model, o = make(d, B=96)
    dups = sum(check_batch(model, o, d, 20141119, ep, u0, 96) for ep, u0 in ((0, 0), (3, 96), (1, 1200 - 96)))
    assert dups > 0
    # a hot item longer than the LDS window of segment_sort_kernel (3072 examples): every user rated item 0
    rng = np.random.default_rng(1)
    rows = [np.unique(np.r_[0, rng.choice(np.arange(1, 400), 30, replace=False)]).astype(np.uint32) for _ in range(3500)]
    ptr = np.r_[0, np.cumsum([r.size for r in rows])].astype(np.int64)
    d2 = synth.Interactions(len(rows), 400, ptr, np.concatenate(rows), np.zeros(len(rows) + 1, np.int64), np.empty(0, np.uint32))
    m2, o2 = make(d2, B=3500, num_neg=1)
    check_batch(m2, o2, d2, 2, 0, 0, 3500, num_neg=1)


def test_bucket_sort_beyond_its_lds_window(built):
    """bucket_sort_kernel's slow paths, on the SHIPPED library: (a) ONE item with more examples than the LDS window holds (4096: every
    one of 7 000 users rated item 0) is ranked in global memory — the range's cells hold more than the window, so the workgroup scans;
    (b) a range whose batch holds far more examples than the ranges were cut for (the first 200 users all rate the same 60 items, the cut
    expects the data set's average): the units' cells overflow (> 31 examples of one unit in one range), sample_kernel raises the batch's
    tag, every workgroup scans, and the hot range is taken in several groups of items.  Same bits as numpy's stable sort."""
    rng = np.random.default_rng(3)
    rows = [np.unique(np.r_[0, rng.choice(np.arange(1, 400), 30, replace=False)]).astype(np.uint32) for _ in range(7000)]
    ptr = np.r_[0, np.cumsum([r.size for r in rows])].astype(np.int64)
    d1 = synth.Interactions(len(rows), 400, ptr, np.concatenate(rows), np.zeros(len(rows) + 1, np.int64), np.empty(0, np.uint32))
    m1, o1 = make(d1, B=7000, num_neg=1)
    check_batch(m1, o1, d1, 2, 0, 0, 7000, num_neg=1)
    rows = [np.arange(60, dtype=np.uint32) if u < 200 else np.sort(rng.choice(np.arange(60, 1000), 20, replace=False)).astype(np.uint32)
            for u in range(2000)]
    ptr = np.r_[0, np.cumsum([r.size for r in rows])].astype(np.int64)
    d2 = synth.Interactions(len(rows), 1000, ptr, np.concatenate(rows), np.zeros(len(rows) + 1, np.int64), np.empty(0, np.uint32))
    m2, o2 = make(d2, B=200, num_neg=3)
    assert check_batch(m2, o2, d2, 5, 0, 0, 200, num_neg=3) > 0          # 12 000 positives on 60 items of one range
    check_batch(m2, o2, d2, 5, 1, 900, 200, num_neg=3)                    # and an ordinary batch of the same handle


def test_tile_sort_with_more_than_16384_items(built, monkeypatch, devlib):
    """per-tile cursors above 64 KiB of LDS (dynamic LDS attribute): 20 000 items"""
    monkeypatch.setenv("CDAE_SORT_TILE", "1")
    d = synth.generate(600, 20_000, 40_000, seed=4)
    model, o = make(d, B=128)
    check_batch(model, o, d, 7, 0, 0, 128)
    check_batch(model, o, d, 7, 1, 472, 128)
