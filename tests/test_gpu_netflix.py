"""-m gpu: BASELINE configs[3] — Netflix-shape synthetic (480 000 users x 17 700 items, ~80 M train interactions), K=200,
negative-sampling=5, CE — on ONE GPU (the 8-GPU form of the config is the driver's to run; what one GPU can certify is the
path itself at this shape: > 2^32 / 12 examples per epoch, 64-bit row pointers, 1 875 batches per epoch, 384 MB of Wu).

  * integer work bit-exact on 256-user windows (masks, negatives, item-major order, segments, duplicate flags) against the
    oracle's draws — get_corrputed_input cdae.hpp:361-371, sample_negative_item recsys_model_base.hpp:46-57
  * `batch_users` = 1 on the leading users against the LITERAL restatement (cdae.hpp:136-146, 198-358), parameters <= 2e-4 of range
  * the committed literal fixtures (tests/golden/netflix_k200_ce_literal_seed*.npz: CPU oracle, strictly sequential, fp64, every
    user trained, Recall@10 over the first `eval_users` users) against the HIP path at the LIBRARY DEFAULT batch_users — the
    handle is created with batch_users = 0, exactly as src/model/recsys/cdae.hpp does without CDAE_BATCH_USERS
"""
import glob
import os

import numpy as np
import pytest

import cdae_amd
from cdae_amd import synth
import oracle as orc
from oracle import binding as ob
from test_gpu_integer import check_batch, make
from test_gpu_accuracy import HYPER, RECALL_TOL_SEED, LOSS_SCHEDULE_OFFSET, LOSS_TOL_AROUND_OFFSET

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "netflix_k200_ce_literal_seed*.npz")))
_data = {}


def data_seed_of(path):
    f = np.load(path, allow_pickle=True)
    return int(f["data_seed"]) if "data_seed" in f.files else int(f["seed"])


_GEN = """
import importlib.util, sys, numpy as np
spec = importlib.util.spec_from_file_location("synth", sys.argv[1])
synth = importlib.util.module_from_spec(spec); sys.modules["synth"] = synth; spec.loader.exec_module(synth)
d = synth.generate_shape("netflix", seed=int(sys.argv[2]))
np.savez(sys.argv[3], train_ptr=d.train_ptr, train_col=d.train_col, test_ptr=d.test_ptr, test_col=d.test_col, shape=np.array([d.num_users, d.num_items]))
"""


def netflix(seed):
    """The Netflix-shape data set of `seed`.  The first call generates the sets of ALL committed fixtures side by side, each in a
    child interpreter that imports nothing but numpy and cdae_amd/synth.py (one generation is ~60-80 s of numpy sorting on one core
    and there is one per data seed: four of them one after the other were most of this module's run time).  The generator and its
    output are the same — only where it runs differs; a box with few cores generates in this process, one set at a time."""
    if not _data:
        seeds = sorted({20141119} | {data_seed_of(p) for p in FIXTURES})
        if len(seeds) > 1 and (os.cpu_count() or 1) >= 2 * len(seeds):
            import subprocess
            import sys
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                procs = [(s, os.path.join(tmp, f"nf{s}.npz"),) for s in seeds]
                running = [(s, out, subprocess.Popen([sys.executable, "-c", _GEN, os.path.join(ROOT, "cdae_amd", "synth.py"), str(s), out]))
                           for s, out in procs]
                for s, out, pr in running:
                    assert pr.wait() == 0, f"generation of the Netflix-shape data set {s} failed"
                    f = np.load(out)
                    _data[s] = synth.Interactions(int(f["shape"][0]), int(f["shape"][1]), f["train_ptr"], f["train_col"], f["test_ptr"], f["test_col"])
    if seed not in _data:
        _data[seed] = synth.generate_shape("netflix", seed=seed)
    return _data[seed]


def test_netflix_shape_integer_work_bit_exact(built):
    d = netflix(20141119)
    assert d.num_users == 480_000 and d.num_items == 17_700 and d.nnz_train > 70_000_000
    model, o = make(d, K=8, B=256)
    dups = 0
    for ep, u0 in ((0, 0), (2, 256 * 911), (1, d.num_users - 256)):
        dups += check_batch(model, o, d, 20141119, ep, u0, 256)
    assert dups > 0
    model.close()


def test_netflix_shape_batch_users_one_is_the_literal_schedule(built):
    d = netflix(20141119)
    n = 1500
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, batch_users=1, **HYPER))
    m.reset(d, seed=5)
    o = orc.Oracle(orc.OracleConfig(num_dim=200, loss_type=ob.LOSS_CE, **HYPER), d.num_users, d.num_items, d.train_ptr, d.train_col)
    for which in (cdae_amd.P_W, cdae_amd.P_W_AG, cdae_amd.P_B, cdae_amd.P_B_AG, cdae_amd.P_BP, cdae_amd.P_BP_AG):
        o.set(which, m.get(which).astype(np.float64))
    wu0 = m.get(cdae_amd.P_WU)
    o.set(ob.P_WU, wu0.astype(np.float64))
    o.set(ob.P_WU_AG, m.get(cdae_amd.P_WU_AG).astype(np.float64))
    st = m.train_users(9, 0, 0, n)
    assert st.users == n and st.batches == n
    o.train_literal(9, 0, 0, n)
    for which, rows in ((cdae_amd.P_W, d.num_items), (cdae_amd.P_W_AG, d.num_items), (cdae_amd.P_BP, 0), (cdae_amd.P_B, 0), (cdae_amd.P_WU, n)):
        got, ref = m.get(which).astype(np.float64), o.get(which)
        ref = ref.reshape(got.shape)
        if which == cdae_amd.P_WU:
            got, ref = got[:n], ref[:n]
        err = np.abs(got - ref).max() / (1e-3 + np.abs(ref).max())
        assert err <= 2e-4, (which, err)
    m.close()


def test_there_is_a_netflix_fixture():
    assert len(FIXTURES) >= 1, "tests/golden/make_literal_curves.py --shape netflix --eval-users 60000"


FIXTURES.sort(key=lambda p: (data_seed_of(p) != 20141119, data_seed_of(p), p))   # one data set = one generation; the first tests' set first
_curves = {}


def curves_of(path):
    if path not in _curves:
        f = np.load(path, allow_pickle=True)
        seed, K, ne = int(f["seed"]), int(f["num_dim"]), int(f["eval_users"])
        assert str(f["shape"]) == "netflix" and K == 200 and str(f["loss"]) == "CE"
        d = netflix(data_seed_of(path))
        assert d.nnz_train == int(f["nnz_train"]), "the synthetic generator changed: regenerate the fixtures"
        m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, batch_users=0, **HYPER))
        m.reset(d, seed=seed)
        import bench
        assert m.batch_users == bench.DEFAULT_BATCH_USERS == 256
        rec, loss = [], []
        for ep in range(len(f["recall10"])):
            st = m.train_one_iteration(seed, ep)
            assert st.users == d.num_users and st.batches == -(-d.num_users // 256)
            loss.append(m.current_loss(seed, ep))
            rec.append(orc.eval_topn(m.recommend_all(10, 0, ne), d.test_ptr[:ne + 1], d.test_col[:d.test_ptr[ne]])[5])
        m.close()
        rec, loss = np.array(rec), np.array(loss)
        print(f"\nnetflix data seed {data_seed_of(path)} stream seed {seed}: recall@10 hip {np.round(rec, 5)} literal {np.round(f['recall10'], 5)} "
              f"d {np.round(rec - f['recall10'], 5)}; loss hip/literal - 1 {np.round(loss / f['train_loss'] - 1, 4)}")
        _curves[path] = (rec, np.asarray(f["recall10"]), loss, np.asarray(f["train_loss"]))
    return _curves[path]


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_netflix_literal_fixture_at_the_library_default_batch_users(built, path):
    rec, ref_rec, loss, ref_loss = curves_of(path)
    assert np.abs(rec - ref_rec).max() <= RECALL_TOL_SEED
    assert np.abs(loss / ref_loss - 1.0 - LOSS_SCHEDULE_OFFSET).max() <= LOSS_TOL_AROUND_OFFSET


def test_netflix_mean_recall_difference_over_the_seeds(built):
    """the mean-over-seeds statement of tests/test_gpu_accuracy.py at Netflix shape, once there are enough fixtures to carry it
    (stream seeds on one data set count: what is averaged out is the schedule's sensitivity to the random streams)"""
    if len(FIXTURES) < 4:
        pytest.skip(f"{len(FIXTURES)} Netflix-shape fixtures: a mean over fewer than four seeds does not resolve 0.0015")
    n = min(len(curves_of(p)[0]) for p in FIXTURES)          # (a fixture still being generated holds fewer epochs)
    d = np.array([(curves_of(p)[0] - curves_of(p)[1])[:n] for p in FIXTURES])
    mean = d.mean(axis=0)
    print(f"\n{len(FIXTURES)} Netflix-shape seeds: mean signed dRecall@10 per epoch {np.round(mean, 5)}, std {np.round(d.std(axis=0, ddof=1), 5)}")
    from test_gpu_accuracy import RECALL_TOL_MEAN
    assert np.abs(mean).max() <= RECALL_TOL_MEAN, mean


# ---- BASELINE configs[3] as it is named: Netflix shape on 8 GPUs, data parallel over users (cdae_hip_multi_set_schedule) ------------------
# Eight logical shards of the one GPU here.  One relayed epoch (the single-GPU schedule handed from shard to shard: exact), then
# synchronous exchanged steps of 64 users per shard, global-accumulator combine.  Round 5 measured on the four fixtures of round 3
# (profiles/r05_schedule_envelope_netflix.txt): per seed <= 0.0046, mean over the seeds -0.0009 / +0.0020 / +0.0009 — at the edge of the north
# star's 0.002, outside the single GPU's 0.0015: the bounds below say so.
_sched = {}


def sched_curves_of(path, shards=8, sync_users=64, relay=1.0):
    if path not in _sched:
        f = np.load(path, allow_pickle=True)
        seed, K, ne = int(f["seed"]), int(f["num_dim"]), int(f["eval_users"])
        d = netflix(data_seed_of(path))
        m = cdae_amd.MultiCDAE(cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, batch_users=256, **HYPER), devices=[0] * shards)
        m.set_schedule(period=0, combine=cdae_amd.COMBINE_GLOBAL_ACC, sync_batch_users=sync_users, relay_epochs=relay)
        m.reset(d, seed=seed)
        rec, loss = [], []
        for ep in range(len(f["recall10"])):
            st = m.train_one_iteration(seed, ep)
            assert st.users == d.num_users
            loss.append(m.current_loss(seed, ep))
            rec.append(orc.eval_topn(m.recommend_all(10, 0, ne), d.test_ptr[:ne + 1], d.test_col[:d.test_ptr[ne]])[5])
        m.close()
        _sched[path] = (np.array(rec), np.asarray(f["recall10"]), np.array(loss), np.asarray(f["train_loss"]))
    return _sched[path]


def test_netflix_relay_then_exchange_schedule_on_eight_shards(built):
    n = min(len(sched_curves_of(p)[0]) for p in FIXTURES)
    d = np.array([(sched_curves_of(p)[0] - sched_curves_of(p)[1])[:n] for p in FIXTURES])
    lo = np.array([(sched_curves_of(p)[2] / sched_curves_of(p)[3] - 1.0)[:n] for p in FIXTURES])
    print(f"\nNetflix shape, 8 user shards x 64 users per step after one relayed epoch, {len(FIXTURES)} seeds: mean signed dRecall@10 per epoch "
          f"{np.round(d.mean(axis=0), 5)}, max |d| {np.round(np.abs(d).max(axis=0), 5)}; loss offset {np.round(lo.mean(axis=0), 4)}")
    assert np.abs(d).max() <= 0.006, np.abs(d).max(axis=0)                 # per seed (measured 0.0046; the single GPU's bound is 0.005)
    assert np.abs(d.mean(axis=0)).max() <= 0.003, d.mean(axis=0)           # mean over the seeds (measured 0.0020; the single GPU's bound is 0.0015)
    assert np.abs(lo[:, 1:]).max() <= 0.01, lo


# ---- a longer horizon at Netflix shape: one seed of the literal schedule beyond the three epochs of the fixtures above ---------------------
# `netflix_k200_ce_literal20_seed*.npz` (make_literal_curves.py --epochs 20 --tag-suffix 20: an epoch of the fp64 loop + its top-10 of
# 60 000 users is ~15 min of one core here; the run is resumable: --checkpoint).
LONG = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "netflix_k200_ce_literal20_seed*.npz")))


@pytest.mark.skipif(not LONG, reason="no long Netflix-shape literal fixture")
def test_netflix_default_schedule_beyond_three_epochs(built):
    f = np.load(LONG[0], allow_pickle=True)
    seed, ne, n = int(f["seed"]), int(f["eval_users"]), len(f["recall10"])
    assert n >= 5, "the long fixture should reach beyond the three-epoch ones"
    d = netflix(data_seed_of(LONG[0]))
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, batch_users=0, **HYPER))
    m.reset(d, seed=seed)
    rec, loss = [], []
    for ep in range(n):
        m.train_one_iteration(seed, ep)
        loss.append(m.current_loss(seed, ep))
        rec.append(orc.eval_topn(m.recommend_all(10, 0, ne), d.test_ptr[:ne + 1], d.test_col[:d.test_ptr[ne]])[5])
    m.close()
    dr, lo = np.array(rec) - f["recall10"], np.array(loss) / f["train_loss"] - 1.0
    print(f"\nNetflix shape, seed {seed}, {n} epochs at the library default: dRecall@10 {np.round(dr, 5)}; loss offset {np.round(lo, 4)}")
    assert np.abs(dr).max() <= RECALL_TOL_SEED, dr                  # one seed: the per-seed bound (the literal schedule's own seed spread)
    assert lo.max() <= 0.005 and lo.min() >= -0.035, lo             # the known schedule offset, not growing
