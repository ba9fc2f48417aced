"""-m gpu: the north star's accuracy bar as a driver-run fact.

"reproducing the reference CPU solver's loss curve and Recall@N within a stated fp tolerance on identical inputs …
Recall@10 within +/-0.002 of reference" (BASELINE.json).  The expected curves are committed fixtures made by
tests/golden/make_literal_curves.py from the CPU oracle's LITERAL schedule — train_one_iteration strictly user by user in
fp64 (cdae.hpp:136-358), Train Loss = data_loss + penalty_loss (solver-inl.hpp:55), Recall@10 = rets[5] of
evaluation.hpp:183-219 — six data/stream seeds at the BASELINE shape (ML-10M-shape 70 000 x 10 600, K=200, neg=5, CE).
The HIP path runs the same data, init and counter-based random streams at **bench.py's default `batch_users`** — the
throughput bench.py reports is only meaningful inside this envelope.

What "within +/-0.002 of the reference" can mean on this data set — measured, not assumed (tools/reference_noise.py,
profiles/r02_reference_noise.txt): the LITERAL schedule's own Recall@10 moves with the random streams alone (initial values,
dropout masks, negatives; same data) by a standard deviation of 0.0011-0.0024 per epoch, 0.0034-0.0064 between the best and
the worst of eight stream seeds.  A single-seed comparison therefore cannot resolve 0.002; the mean over seeds can.  The same
eight streams at batch_users 256 differ from their literal twins by 0.0008-0.0015 (std, paired), with a mean of +0.0019 at
epoch 1 (the batched schedule is slightly AHEAD after one epoch, 3.5 standard errors) and |mean| <= 0.0004 at epochs 2-5
(not distinguishable from zero); a second data set with 24 streams: literal best - worst 0.0073-0.0102, paired mean
+0.0001 / -0.0002 / +0.0002 / +0.0003 / +0.0009.

Stated tolerances (each asserted below; six data/stream seeds):
  * per seed, every epoch: |Recall@10_hip - Recall@10_literal| <= 0.005 — the literal schedule's own best-to-worst spread over
    stream seeds; anything larger is not seed noise
  * MEAN over the seeds of the signed difference: |mean| <= 0.0015 at EVERY epoch — inside the north star's 0.002 and about 2.5
    standard errors of a six-seed mean (~0.0006).  Measured on the six fixtures: +0.0010 / -0.0007 / -0.0008 / -0.0009 / -0.0012.
    The north star's "+-0.002 of reference" is therefore certified as a MEAN-OVER-SEEDS statement; a single run can differ by
    up to the reference's own seed noise (first bullet)
  * reported train loss: the batched schedule reads systematically LOW (-1.6 ... -2.6 % at 256 users per batch, every seed and
    epoch; -1.1 % once, at the first epoch of seed 99); asserted as a band around that known offset, |loss/literal - 1 + 0.019| <=
    0.011, so neither a loss above the literal one nor a further 1 % of offset passes unnoticed; and the curve has the same shape: the epoch-to-epoch change agrees in sign
    wherever the literal curve moves by more than 0.5 %
  * `batch_users` = 1 IS the reference schedule: one full-size epoch reproduces the fixture's Recall@10 to 1e-4 and its loss
    to 2e-4 relative (fp32 device arithmetic against the fp64 oracle over 70 000 sequential users)
Round-3: the mean bound went 0.002 / 0.003 (epoch 1) -> 0.0015 everywhere and the loss bound from +-3 % to the band above (review).
Round-2 history: with three seeds the per-seed bound was 0.002 final / 0.003 per epoch and all three passed; three more seeds
(42, 99, 314159) gave |d| up to 0.0046 (seed 99, epoch 1) and 0.0037 (seed 314159, final) — which is what prompted measuring
the reference's own noise above.  Batch sizes: the sweep in DESIGN.md §2 (tools/accuracy_envelope.py) is flat in the batch
size up to 256 (mean |d| ~0.001) and shows a systematic offset from 320 on (mean |d| 0.002, max 0.005 at 384-512, always
towards lower Recall), so bench.py's default is 256.  The loss offset is systematic and linear in the batch size (-0.3 % at
16, -1.3 % at 128, -2.2 % at 256, -4 % at 512): the hidden layer of a batch is evaluated against the batch-start snapshot
(DESIGN.md §2), so it is a schedule tolerance, not fp noise (fp32-vs-fp64 alone: the batch_users = 1 test).
"""
import glob
import os

import numpy as np
import pytest

import cdae_amd
from cdae_amd import synth
import oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECALL_TOL_SEED = 0.005           # per seed and epoch: the literal schedule's own spread over stream seeds
RECALL_TOL_MEAN = 0.0015          # mean over the six seeds, every epoch: inside the north star's 0.002, ~2.5 standard errors (0.0006)
# Train loss: the 256-user schedule reads 1.6-2.6 % LOW at every seed and epoch (the hidden layer of a batch is evaluated against
# the batch-start snapshot) — a known, systematic schedule offset, not a tolerance to hide drift in: the bound is a band
# AROUND it (measured over the six seeds x 5 epochs: -0.0264 ... -0.0114, the first epoch being the most variable), not 3 %
# either way.
LOSS_SCHEDULE_OFFSET = -0.019
LOSS_TOL_AROUND_OFFSET = 0.011
HYPER = dict(num_neg=5, num_corruptions=1, corruption_ratio=0.5, scaled=True, learn_rate=0.1, beta=1.0, lambda_=0.01)

FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ml10m_k200_ce_literal_seed*.npz")))
_curves = {}                       # fixture path -> (recall hip, recall literal, loss hip, loss literal): trained once per session


def bench_default_batch_users():
    import bench
    return bench.DEFAULT_BATCH_USERS


def curves_of(path):
    if path not in _curves:
        f = np.load(path, allow_pickle=True)
        seed, K = int(f["seed"]), int(f["num_dim"])
        assert str(f["shape"]) == "ml10m" and K == 200 and str(f["loss"]) == "CE" and int(f["full_output_batch"]) == 0
        ref_rec, ref_loss = f["recall10"], f["train_loss"]
        d = synth.generate_shape("ml10m", seed=seed)
        assert d.nnz_train == int(f["nnz_train"]), "the synthetic generator changed: regenerate the fixtures"
        B = bench_default_batch_users()
        m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, batch_users=B, **HYPER))
        m.reset(d, seed=seed)
        rec, loss = [], []
        for ep in range(len(ref_rec)):
            m.train_one_iteration(seed, ep)
            loss.append(m.current_loss(seed, ep))
            rec.append(orc.eval_topn(m.recommend_all(10), d.test_ptr, d.test_col)[5])
        m.close()
        rec, loss = np.array(rec), np.array(loss)
        print(f"\nseed {seed} batch_users {B}\n  recall@10 hip     {np.round(rec, 5)}\n  recall@10 literal {np.round(ref_rec, 5)}"
              f"\n  d                 {np.round(rec - ref_rec, 5)}\n  loss hip/literal - 1 {np.round(loss / ref_loss - 1, 4)}")
        _curves[path] = (rec, np.asarray(ref_rec), loss, np.asarray(ref_loss))
    return _curves[path]


def test_there_are_at_least_six_seeds():
    assert len(FIXTURES) >= 6, FIXTURES


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_recall_and_loss_curve_at_bench_batch_users(built, path):
    rec, ref_rec, loss, ref_loss = curves_of(path)
    assert np.abs(rec - ref_rec).max() <= RECALL_TOL_SEED, (path, rec - ref_rec)
    assert np.abs(loss / ref_loss - 1.0 - LOSS_SCHEDULE_OFFSET).max() <= LOSS_TOL_AROUND_OFFSET, (path, loss / ref_loss - 1.0)
    moves = np.abs(np.diff(ref_loss)) > 0.005 * ref_loss[:-1]
    assert (np.sign(np.diff(loss))[moves] == np.sign(np.diff(ref_loss))[moves]).all()


def test_mean_recall_difference_over_the_seeds(built):
    """the statement the north star's 0.002 can be held to: no systematic Recall@10 offset against the literal schedule"""
    d = np.array([curves_of(p)[0] - curves_of(p)[1] for p in FIXTURES])        # [seed][epoch], signed
    mean = d.mean(axis=0)
    print(f"\n{len(FIXTURES)} seeds: mean signed dRecall@10 per epoch {np.round(mean, 5)}, std {np.round(d.std(axis=0, ddof=1), 5)}, "
          f"max |d| {np.round(np.abs(d).max(axis=0), 5)}")
    assert np.abs(mean).max() <= RECALL_TOL_MEAN, mean


def test_batch_users_one_is_the_reference_schedule_at_full_size(built):
    """One epoch of 70 000 strictly sequential users on the device (fp32) against the fp64 literal fixture."""
    f = np.load(FIXTURES[0], allow_pickle=True)
    seed = int(f["seed"])
    d = synth.generate_shape("ml10m", seed=seed)
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, batch_users=1, **HYPER))
    m.reset(d, seed=seed)
    m.train_one_iteration(seed, 0)
    loss = m.current_loss(seed, 0)
    rec = orc.eval_topn(m.recommend_all(10), d.test_ptr, d.test_col)[5]
    m.close()
    print(f"\nbatch_users 1, seed {seed}: recall@10 {rec:.6f} vs literal {f['recall10'][0]:.6f}; loss ratio - 1 = {loss / f['train_loss'][0] - 1:.2e}")
    assert abs(rec - f["recall10"][0]) <= 1e-4
    assert abs(loss / f["train_loss"][0] - 1.0) <= 2e-4     # fp32 sum of 8 M positive-example losses (tests/test_gpu_parity.py: same bound)


# ---- the app's own horizon: Solver<CDAE>(model, 50) (/root/reference/apps/yelp/yelp.cpp:197, solver-inl.hpp:72-74) -----------------
# `ml10m_k200_ce_literal50_seed*.npz`: the LITERAL schedule (fp64 oracle, strictly sequential) over 50 epochs, three data / stream seeds
# (tests/golden/make_literal_curves.py --epochs 50 --tag-suffix 50: ~6 min of one core per epoch).  Round 4's fixtures stop at epoch 5,
# where the mean difference of the 256-user schedule was -0.0012 and still falling; these say where it goes.
LONG_FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ml10m_k200_ce_literal50_seed*.npz")))
LONG_MIN_EPOCHS = 20                 # a fixture cut short (the generator saves after every epoch) still counts from here on
RECALL_TOL_MEAN_LONG = 0.002         # the north star's tolerance, as a mean over the seeds, at EVERY epoch up to the last
_long_curves = {}


def long_curves_of(path):
    if path not in _long_curves:
        f = np.load(path, allow_pickle=True)
        seed, K = int(f["seed"]), int(f["num_dim"])
        ref_rec, ref_loss = np.asarray(f["recall10"]), np.asarray(f["train_loss"])
        d = synth.generate_shape("ml10m", seed=seed)
        assert d.nnz_train == int(f["nnz_train"]) and K == 200
        m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, batch_users=bench_default_batch_users(), **HYPER))
        m.reset(d, seed=seed)
        m.set_test_rows(d.test_ptr, d.test_col)        # TOPN on the device (bit-equal to the oracle's evaluator: tests/test_gpu_eval.py)
        rec, loss = [], []
        for ep in range(len(ref_rec)):
            m.train_one_iteration(seed, ep)
            loss.append(m.current_loss(seed, ep))
            rec.append(m.eval_topn(10)[0][5])
        m.close()
        _long_curves[path] = (np.array(rec), ref_rec, np.array(loss), ref_loss)
    return _long_curves[path]


def test_there_are_long_horizon_fixtures():
    assert len(LONG_FIXTURES) >= 3, LONG_FIXTURES
    for p in LONG_FIXTURES:
        assert len(np.load(p, allow_pickle=True)["recall10"]) >= LONG_MIN_EPOCHS, p


def test_recall_and_loss_over_the_apps_own_horizon(built):
    """bench.py's default batch_users against the literal schedule for as many epochs as the app trains (50): the mean over the seeds of
    the signed Recall@10 difference stays inside the north star's +-0.002 at EVERY epoch including the last; a single seed stays inside
    the literal schedule's own seed-to-seed spread; the reported loss stays in the band around the schedule's known offset."""
    n = min(len(np.load(p, allow_pickle=True)["recall10"]) for p in LONG_FIXTURES)
    d = np.array([long_curves_of(p)[0][:n] - long_curves_of(p)[1][:n] for p in LONG_FIXTURES])
    lo = np.array([long_curves_of(p)[2][:n] / long_curves_of(p)[3][:n] - 1.0 for p in LONG_FIXTURES])
    mean = d.mean(axis=0)
    idx = [e for e in (0, 4, 9, 19, 29, 39, 49) if e < n]
    print(f"\n{len(LONG_FIXTURES)} seeds x {n} epochs at batch_users {bench_default_batch_users()}: mean signed dRecall@10 at epochs "
          f"{[e + 1 for e in idx]}: {np.round(mean[idx], 5)}; max |mean| {np.abs(mean).max():.5f} (epoch {int(np.abs(mean).argmax()) + 1}); "
          f"max |d| per seed {np.round(np.abs(d).max(axis=1), 5)}; literal Recall@10 at the last epoch {np.round([long_curves_of(p)[1][n - 1] for p in LONG_FIXTURES], 4)}; "
          f"loss offset {lo.min():.4f} ... {lo.max():.4f} (last epoch {np.round(lo[:, -1], 4)})")
    assert np.abs(mean).max() <= RECALL_TOL_MEAN_LONG, mean
    assert np.abs(d).max() <= 0.007, np.abs(d).max(axis=0)             # (measured: see DESIGN.md §2; the literal schedule's own best - worst over stream seeds is 0.0034-0.0102)
    assert lo.max() <= 0.005 and lo.min() >= -0.035, (lo.min(), lo.max())  # the batched schedule reads LOW (-2.3 % at epoch 1, shrinking to -1 % by epoch 20): the offset does not grow


# ---- the user-sharded schedule of cdae_hip_multi_set_schedule: relay warm-up, then exchanged steps ---------------------
# Eight logical shards of the one GPU a test box has; the first epoch on the single-GPU schedule handed from shard to shard (exact), the
# rest as synchronous exchanged steps of 64 users per shard (512 per global step) folded in by the global-accumulator rule.  Round 5
# measured (profiles/r05_schedule_envelope_*.txt): after the relay the mean over six seeds stays within the north star's +-0.002, but a
# single seed moves by up to 0.0086 — more than the single GPU's 0.005: the bounds below are THIS schedule's, not the single GPU's.
SCHED_SHARDS, SCHED_SYNC_USERS, SCHED_RELAY = 8, 64, 1.0
_sched_curves = {}


def sched_curves_of(path):
    if path not in _sched_curves:
        f = np.load(path, allow_pickle=True)
        seed, K = int(f["seed"]), int(f["num_dim"])
        d = synth.generate_shape("ml10m", seed=seed)
        m = cdae_amd.MultiCDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, batch_users=bench_default_batch_users(), **HYPER), devices=[0] * SCHED_SHARDS)
        m.set_schedule(period=0, combine=cdae_amd.COMBINE_GLOBAL_ACC, sync_batch_users=SCHED_SYNC_USERS, relay_epochs=SCHED_RELAY)
        m.reset(d, seed=seed)
        rec, loss = [], []
        for ep in range(len(f["recall10"])):
            m.train_one_iteration(seed, ep)
            loss.append(m.current_loss(seed, ep))
            rec.append(m.eval_topn(d.test_ptr, d.test_col, 10)[0][5])
        m.close()
        _sched_curves[path] = (np.array(rec), np.asarray(f["recall10"]), np.array(loss), np.asarray(f["train_loss"]))
    return _sched_curves[path]


def test_relay_then_exchange_schedule_on_eight_shards_at_ml10m_shape(built):
    d = np.array([sched_curves_of(p)[0] - sched_curves_of(p)[1] for p in FIXTURES])
    lo = np.array([sched_curves_of(p)[2] / sched_curves_of(p)[3] - 1.0 for p in FIXTURES])
    print(f"\n{SCHED_SHARDS} user shards x {SCHED_SYNC_USERS} users per step after {SCHED_RELAY} relayed epoch(s), {len(FIXTURES)} seeds: mean signed dRecall@10 per epoch "
          f"{np.round(d.mean(axis=0), 5)}, max |d| {np.round(np.abs(d).max(axis=0), 5)}; loss offset per epoch {np.round(lo.mean(axis=0), 4)}")
    # the relayed epoch IS the single-GPU schedule: its bounds
    assert np.abs(d[:, 0]).max() <= RECALL_TOL_SEED and abs(d[:, 0].mean()) <= RECALL_TOL_MEAN
    # the exchanged epochs: mean over the seeds inside the north star's 0.002 (measured <= 0.0014), a single seed within 0.010 (measured 0.0086)
    assert np.abs(d[:, 1:].mean(axis=0)).max() <= 0.002, d.mean(axis=0)
    assert np.abs(d[:, 1:]).max() <= 0.010, np.abs(d).max(axis=0)
    assert np.abs(lo[:, 1:]).max() <= 0.015, lo                            # (measured -0.003 ... -0.006: closer to the literal loss than the single GPU's -0.019)


# ---- the multi-GPU schedule with an accuracy claim: sampled decode in the item-rows layout ------------------
# It is the single-GPU schedule over item shards (per-row chains sequential over the GLOBAL batch, two all-reduced per-user sums),
# so it must hold the SAME bounds as the single handle; run here as four logical shards of the one GPU a test box has.
_shard_curves = {}


def shard_curves_of(path, shards=4):
    if path not in _shard_curves:
        f = np.load(path, allow_pickle=True)
        seed, K = int(f["seed"]), int(f["num_dim"])
        d = synth.generate_shape("ml10m", seed=seed)
        B = bench_default_batch_users()
        m = cdae_amd.MultiCDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, batch_users=B, **HYPER), devices=[0] * shards, item_rows=True)
        m.reset(d, seed=seed)
        rec, loss = [], []
        for ep in range(len(f["recall10"])):
            m.train_one_iteration(seed, ep)
            loss.append(m.current_loss(seed, ep))
            rec.append(orc.eval_topn(m.recommend_all(10), d.test_ptr, d.test_col)[5])
        m.close()
        _shard_curves[path] = (np.array(rec), np.asarray(f["recall10"]), np.array(loss), np.asarray(f["train_loss"]))
    return _shard_curves[path]


def test_item_rows_sampled_layout_holds_the_accuracy_bounds_at_ml10m_shape(built):
    d = np.array([shard_curves_of(p)[0] - shard_curves_of(p)[1] for p in FIXTURES])
    lo = np.array([shard_curves_of(p)[2] / shard_curves_of(p)[3] - 1.0 for p in FIXTURES])
    single = np.array([curves_of(p)[0] for p in FIXTURES])
    sharded = np.array([shard_curves_of(p)[0] for p in FIXTURES])
    print(f"\n4 item shards, {len(FIXTURES)} seeds: mean signed dRecall@10 per epoch {np.round(d.mean(axis=0), 5)}, max |d| {np.round(np.abs(d).max(axis=0), 5)}; "
          f"loss offset {np.round(lo.min(), 4)} ... {np.round(lo.max(), 4)}; max |Recall@10 sharded - single GPU| {np.abs(sharded - single).max():.5f}")
    assert np.abs(d).max() <= RECALL_TOL_SEED
    assert np.abs(d.mean(axis=0)).max() <= RECALL_TOL_MEAN
    assert np.abs(lo - LOSS_SCHEDULE_OFFSET).max() <= LOSS_TOL_AROUND_OFFSET
    # and it is the SAME trajectory as the single GPU's up to fp32 association of two sums: Recall@10 moves by a few users' lists at most
    assert np.abs(sharded - single).max() <= 5e-4


# ---- BASELINE configs[1]: Yelp-shape K=50 FULL-OUTPUT decode (bf16 MFMA), CE ---------------------------------------
# The reference has no full-output training.  Two anchors, both committed fp64 fixtures of the oracle:
#   * `yelp_k50_ce_full512_seed*.npz` — the BLOCK schedule at bench.py's block size (Oracle.train_full: every unrated item a negative
#     with target 0, one summed AdaGrad step per decoder row per block of 512 users), 40 epochs: what the HIP kernels must compute.
#     The HIP path rounds Z, D and g to bf16 for the three products (fp32 accumulate), so the tolerance is a bf16 one.
#   * `yelp_k50_ce_full1_seed*.npz` — the B = 1 LIMIT, i.e. the reference's own loop (cdae.hpp:225-293) fed every unrated item, 30
#     epochs: what "the reference's semantics" means for this north-star extension (tests/test_oracle.py pins full-output(1) == the
#     literal step fed all unrated items).
# Round 4 measured what relates them (tools/accuracy_envelope.py --full-output, DESIGN.md §5c, four seeds): NO block size above 1
# follows the literal loop's trajectory — a block takes ONE step per decoder row where the loop takes B, so at equal epochs a larger
# block is behind early (epoch 5: literal 0.154, B = 64 0.171, B = 512 0.065) and every block size ends ABOVE it (the literal loop
# plateaus at Recall@10 0.20 after 30 epochs; blocks of 16 ... 256 reach 0.26, 512 reaches 0.254 at epoch 40 and is still rising).
# The claim the bench block size can be held to is therefore ONE-SIDED and a mean over seeds: the block schedule reaches the literal
# loop's 30-epoch best Recall@10 within a stated number of epochs and stays above it.
FULL_FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "yelp_k50_ce_full512_seed*.npz")))
FULL_LITERAL = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "yelp_k50_ce_full1_seed*.npz")))
FULL_BENCH_BLOCK = 512                     # bench.py --full-output --shape yelp --num-dim 50 runs this block size (configs[1])
FULL_EPOCHS_TO_LITERAL_BEST = 28           # measured 24 / 26 / 27 / 26 on the four seeds (fp64 oracle) — asserted per seed with two epochs of slack
_full_curves = {}


def full_curves_of(path):
    if path not in _full_curves:
        f = np.load(path, allow_pickle=True)
        seed, K, B = int(f["seed"]), int(f["num_dim"]), int(f["full_output_batch"])
        assert K == 50 and B == FULL_BENCH_BLOCK
        d = synth.generate_shape("yelp", seed=seed)
        assert d.nnz_train == int(f["nnz_train"])
        m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, batch_users=B, full_output=True, **HYPER))
        m.reset(d, seed=seed)
        m.set_test_rows(d.test_ptr, d.test_col)
        rec, loss, probes = [], [], None
        for ep in range(len(f["recall10"])):
            m.train_one_iteration(seed, ep)
            loss.append(m.current_loss(seed, ep))
            rec.append(m.eval_topn(10)[0][5])
        W = m.get(cdae_amd.P_W).astype(np.float64)
        probes = (np.abs(W[f["probe_items"]] - f["W_rows"]).max() / float(f["W_absmax"]),
                  np.abs(m.get(cdae_amd.P_B) - f["b"]).max() / max(1e-3, np.abs(f["b"]).max()))
        m.close()
        _full_curves[path] = (np.array(rec), np.asarray(f["recall10"]), np.array(loss), np.asarray(f["train_loss"]), probes, seed)
    return _full_curves[path]


@pytest.mark.parametrize("path", FULL_FIXTURES, ids=[os.path.basename(p)[:-4] for p in FULL_FIXTURES])
def test_yelp_shape_full_output_k50_curve(built, path):
    """schedule parity: the HIP block schedule IS the oracle's block schedule at the bench block size, epoch by epoch over 40 epochs"""
    rec, ref_rec, loss, ref_loss, probes, seed = full_curves_of(path)
    print(f"\nseed {seed}: max |recall@10 hip - oracle| {np.abs(rec - ref_rec).max():.5f} (first 10 epochs {np.abs(rec - ref_rec)[:10].max():.5f}); "
          f"max |loss hip/oracle - 1| {np.abs(loss / ref_loss - 1).max():.5f}; probes W {probes[0]:.4f} b {probes[1]:.4f}")
    # the first epochs: the bf16 tolerance of round 2 (0.003 / 1 %); over 40 epochs the rounding of 800 block steps accumulates and the
    # steep part of the curve (epochs 20-35, +0.015 Recall@10 per epoch) turns a small lag into a visible difference: 0.01 / 1 %
    assert np.abs(rec - ref_rec)[:10].max() <= 0.003
    assert np.abs(rec - ref_rec).max() <= 0.01
    assert np.abs(loss / ref_loss - 1.0).max() <= 0.01
    # (parameters at the probes after 40 epochs = 800 block steps on bf16 operands against fp64: measured up to 0.048 of the range on W;
    # the curves above are the claim, this only guards against a gross divergence)
    from helpers import record_measured
    record_measured(f"full_output_40_epochs_probes_seed{seed}", W=probes[0], b=probes[1])
    assert probes[0] <= 8e-2 and probes[1] <= 1.1e-2          # measured over the four seeds (round 6): W up to 7.6e-2, b up to 8.5e-3 (b's bound was 8e-2)


def test_full_output_block_schedule_reaches_the_literal_loops_quality(built):
    """the accuracy claim of the full-output bench line (config.accuracy): at the bench block size the mean-over-seeds Recall@10 is at
    or above the literal B = 1 loop's 30-epoch best from epoch 28 on (one-sided, +-0.002 on the mean), every seed reaches ITS literal
    twin's best within 28 epochs, and the block schedule never falls back below it afterwards"""
    assert len(FULL_LITERAL) >= 4 and len(FULL_FIXTURES) >= 4
    lit = {int(np.load(p, allow_pickle=True)["seed"]): np.load(p, allow_pickle=True)["recall10"] for p in FULL_LITERAL}
    rows = []
    for p in FULL_FIXTURES:
        rec, _, _, _, _, seed = full_curves_of(p)
        assert seed in lit, seed
        best = float(lit[seed].max())
        reach = next(i + 1 for i, r in enumerate(rec) if r >= best)
        rows.append((seed, best, reach, rec))
        assert reach <= FULL_EPOCHS_TO_LITERAL_BEST + 2, (seed, reach)
        assert (rec[reach - 1:] >= best - 0.002).all(), seed
    mean_rec = np.mean([r[3] for r in rows], axis=0)
    mean_best = float(np.mean([r[1] for r in rows]))
    print(f"\nfull-output, {FULL_BENCH_BLOCK} users per block, {len(rows)} seeds: literal loop's 30-epoch best Recall@10 {mean_best:.4f} (mean); block schedule "
          f"epochs 10 / 20 / 28 / 30 / 40: {np.round(mean_rec[[9, 19, 27, 29, 39]], 4)}; epochs to the literal best per seed {[r[2] for r in rows]}")
    assert (mean_rec[FULL_EPOCHS_TO_LITERAL_BEST - 1:] >= mean_best - 0.002).all()
    assert mean_rec[-1] >= mean_best + 0.03          # and ends well above: 0.254 against 0.201


def test_k512_block_schedule_reaches_the_device_literal_loops_quality(built):
    """The same one-sided statement on configs[4]'s LAUNCHES (K = 512 over 32 768 items: GEMM 1 with the z rows in registers, GEMM 2
    through the transposing read, GEMM 3 fused with the row step), 16 384 users, one seed.  The literal loop here is the HIP path at
    batch_users = 1 — an fp64 CPU epoch of this shape is ~2 h of one core; at Yelp shape the device's batch_users = 1 reproduces the fp64
    literal fixture's Recall@10 to 1e-4 (profiles/r04_full_output_envelope_yelp.txt).  Blocks of 256 users (64 block steps per epoch)
    must be at or above the loop's ten-epoch best after ten epochs — measured 0.1597 against 0.1485; DESIGN.md §5c has four seeds x 20
    epochs x six block sizes."""
    seed, K, EP = 20141119, 512, 10
    d = synth.generate_shape("cfg5_env", seed=seed)

    def curve(B):
        m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, batch_users=B, full_output=True, **HYPER))
        m.reset(d, seed=seed)
        assert m.full_output_plan == (cdae_amd.binding.PLAN_GEMM2_TN | cdae_amd.binding.PLAN_ROWS_FUSED) or B == 1
        m.set_test_rows(d.test_ptr, d.test_col)
        rec = []
        for ep in range(EP):
            m.train_one_iteration(seed, ep)
            rec.append(m.eval_topn(10)[0][5])
        m.close()
        return np.array(rec)

    lit, blk = curve(1), curve(256)
    print(f"\nK=512 x 32768 items, seed {seed}: literal loop (device, batch_users 1) {np.round(lit, 4)}; 256 users per block {np.round(blk, 4)}")
    assert 0.12 < lit.max() < 0.18                       # the envelope's curve (0.1485 at epoch 9), not a degenerate run
    assert blk[-1] >= lit.max() - 0.002
    assert blk[0] < lit[0]                               # ... and it is NOT the loop's trajectory: behind at epoch 1 (0.062 against 0.107)


# ---- ML-10M shape K=200, the second full-output bench line (2048 users per block) --------------------------------------------------
# Anchors: `ml10m_k200_ce_full1_seed*.npz` — the fp64 LITERAL loop (B = 1, every unrated item a negative) for four epochs (~20 min of
# one core per epoch; make_literal_curves.py --shape ml10m --num-dim 200 --full-output-literal), and the device's own B = 1 curve over
# 25 epochs in profiles/r04_full_output_envelope_ml10m.txt (builder-run: 70 000 block steps of ~0.1 ms per epoch).
ML10M_LITERAL = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ml10m_k200_ce_full1_seed*.npz")))
ML10M_ENVELOPE = os.path.join(ROOT, "profiles", "r04_full_output_envelope_ml10m.txt")


def _ml10m_full_curve(seed, B, epochs):
    d = synth.generate_shape("ml10m", seed=seed)
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, batch_users=B, full_output=True, **HYPER))
    m.reset(d, seed=seed)
    m.set_test_rows(d.test_ptr, d.test_col)
    rec, loss = [], []
    for ep in range(epochs):
        m.train_one_iteration(seed, ep)
        if B == 1:
            loss.append(m.current_loss(seed, ep))
        rec.append(m.eval_topn(10)[0][5])
    m.close()
    return np.array(rec), np.array(loss), d


def test_ml10m_shape_device_literal_loop_is_the_fp64_literal_loop(built):
    """batch_users = 1 of the full-output HIP path (bf16 operands) against the fp64 restatement of the reference loop fed every unrated
    item, 70 000 users x 10 600 items x K = 200, two epochs: Recall@10 within 5e-4 (measured 1e-4 / 2.3e-4 over four epochs on the two
    seeds), train loss within 3e-4 relative (measured 3e-5 / 1.1e-4) — so the device's B = 1 curve is what the envelope may call "the literal loop" here too"""
    assert len(ML10M_LITERAL) >= 2
    f = np.load(ML10M_LITERAL[0], allow_pickle=True)
    seed = int(f["seed"])
    rec, loss, d = _ml10m_full_curve(seed, 1, 2)
    assert d.nnz_train == int(f["nnz_train"])
    print(f"\nML-10M shape, seed {seed}: device literal {np.round(rec, 5)} fp64 literal {np.round(f['recall10'][:2], 5)}; loss ratio {loss / f['train_loss'][:2]}")
    assert np.abs(rec - f["recall10"][:2]).max() <= 5e-4
    assert np.abs(loss / f["train_loss"][:2] - 1.0).max() <= 3e-4


def test_ml10m_shape_bench_block_reaches_the_literal_loops_quality(built):
    """the accuracy statement of `bench.py --full-output --batch-users 2048` (config.accuracy): blocks of 2048 users reach the literal
    loop's 25-epoch best Recall@10 within 16 + 2 epochs and stay above it through epoch 25 — one-sided, per seed, all FOUR seeds the
    envelope file holds literal curves for (two through round 4; the review asked for four at the block size the bench line runs).  The
    literal 25-epoch curves are the device's batch_users = 1 runs recorded in profiles/r04_full_output_envelope_ml10m.txt (tied to the
    fp64 loop by the test above); DESIGN.md §5c has four seeds x six block sizes, and the blocks of <= 512 users that pass the loop
    and then over-train"""
    import json
    lit = {}
    for line in open(ML10M_ENVELOPE):
        r = json.loads(line)
        if r.get("run") == "full-output literal":
            lit[int(r["seed"])] = np.array(r["recall10"])
    assert len(lit) >= 4
    for seed in (20141119, 7, 1234, 42):
        best = float(lit[seed].max())
        rec, _, _ = _ml10m_full_curve(seed, 2048, 25)
        reach = next((i + 1 for i, r in enumerate(rec) if r >= best), None)
        print(f"\nML-10M shape, seed {seed}: literal best {best:.4f}; 2048 users per block epochs 5 / 10 / 15 / 20 / 25 {np.round(rec[[4, 9, 14, 19, 24]], 4)}; reached at {reach}")
        assert reach is not None and reach <= 18, (seed, reach)
        assert (rec[reach - 1:] >= best - 0.002).all(), seed
        assert rec[-1] >= best + 0.005                   # 0.170-0.172 against 0.158-0.164
        assert rec[4] < lit[seed][4] - 0.05              # ... and NOT the loop's trajectory: far behind at epoch 5 (0.067 against 0.15)


# ---- reduced BASELINE configs[4]: K=512 full-output over > 65 536 items (three-GEMM path, 256-row tiles, 32-bit keys) --
CFG5_FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cfg5_small_k512_ce_full128_seed*.npz")))


@pytest.mark.parametrize("path", CFG5_FIXTURES, ids=[os.path.basename(p)[:-4] for p in CFG5_FIXTURES])
def test_reduced_config5_k512_131072_items(built, path):
    f = np.load(path, allow_pickle=True)
    seed, K, B = int(f["seed"]), int(f["num_dim"]), int(f["full_output_batch"])
    d = synth.generate_shape("cfg5_small", seed=seed)
    assert d.num_items == 131_072 and K == 512 and d.nnz_train == int(f["nnz_train"])
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, batch_users=B, full_output=True, **HYPER))
    m.reset(d, seed=seed)
    loss = []
    for ep in range(len(f["train_loss"])):
        m.train_one_iteration(seed, ep)
        loss.append(m.current_loss(seed, ep))
    W = m.get(cdae_amd.P_W).astype(np.float64)
    Wu = m.get(cdae_amd.P_WU).astype(np.float64)
    bp = m.get(cdae_amd.P_BP).astype(np.float64)
    errs = dict(W=np.abs(W[f["probe_items"]] - f["W_rows"]).max() / float(f["W_absmax"]),
                W_mean=np.abs(W[f["probe_items"]] - f["W_rows"]).mean() / float(f["W_absmax"]),
                Wu=np.abs(Wu[f["probe_users"]] - f["Wu_rows"]).max() / float(f["Wu_absmax"]),
                bp=np.abs(bp[f["probe_items"]] - f["bp_rows"]).max() / max(1e-3, float(f["bp_absmax"])),
                b=np.abs(m.get(cdae_amd.P_B) - f["b"]).max() / max(1e-3, np.abs(f["b"]).max()),
                loss=abs(loss[-1] / f["train_loss"][-1] - 1.0))
    print("\nreduced config 5:", {k: round(float(v), 5) for k, v in errs.items()})
    # bf16 operands at K = 512 (tests/test_gpu_parity.py: 3e-2 of the range).  W after its first two block steps from a 1e-4
    # accumulator is the worst case: a step is lr * grad / (beta + |grad|), so elements whose block-summed gradient is near 0
    # turn the bf16 rounding of g (2^-9 of ~32 = 0.06) into 0.006 of step each; measured max 4.0e-2, mean < 1e-2 of the range
    from helpers import record_measured
    record_measured("reduced_config5", **errs)
    # measured (round 6, profiles/r06_measured_bf16_guards.txt): Wu 1.06e-2, b' 5.9e-3, b 1.24e-2, W max 3.98e-2, W mean 5.2e-5; bounds <= 1.3 x (3e-2 / 6e-2 / 1e-2 before)
    assert max(errs["Wu"], errs["bp"], errs["b"]) <= 1.6e-2 and errs["W"] <= 5.2e-2 and errs["W_mean"] <= 7e-5, errs
    assert errs["loss"] <= 0.01, errs
    # evaluation at this size goes through the general recommend path (K > 256, 131 072 x 4 B of scores > LDS)
    rec = m.recommend_all(10)
    assert rec.shape == (d.num_users, 10) and rec.max() < d.num_items
    m.close()
