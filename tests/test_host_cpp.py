"""The host C++ layer (src/): libcf-shaped headers over the C ABI.

CPU: the headers compile; the reference's apps/yelp/yelp.cpp builds UNMODIFIED and IN PLACE against them
(only where /root/reference exists — nothing of it is copied here) and runs BASELINE config 1's plumbing
prepare -> split -> test up to the Popularity row.
GPU: the same binaries train CDAE through Solver<CDAE> on the device.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from cdae_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")
REF_YELP = "/root/reference/apps/yelp/yelp.cpp"


def write_ratings(path, seed=5):
    d = synth.generate(300, 120, 9000, seed=seed)
    rng = np.random.default_rng(0)
    pairs = []
    for u in range(d.num_users):
        for ptr, col in ((d.train_ptr, d.train_col), (d.test_ptr, d.test_col)):
            pairs += [(u, int(i)) for i in col[ptr[u]:ptr[u + 1]]]
    rng.shuffle(pairs)
    with open(path, "w") as f:
        f.write("user item\n")
        for u, i in pairs:
            f.write(f"u{u} i{i}\n")
    return len(pairs)


def run(cmd, cwd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    return p.returncode, p.stdout


@pytest.fixture(scope="module")
def host_bins(built):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "src"), "-s", "check"])
    if os.path.exists(REF_YELP):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "src"), "-s", "yelp"])
    return BUILD


def test_host_layer_compiles_and_cpu_pieces_work(host_bins, tmp_path):
    n = write_ratings(tmp_path / "ratings.txt")
    rc, out = run([os.path.join(host_bins, "host_check"), f"--input_file={tmp_path / 'ratings.txt'}"], tmp_path)
    assert rc == 0, out
    assert f"Num of Instance: {n}" in out and "host layer OK" in out
    assert re.search(r"\|\s*P@10\|", out) and "Popularity Model" in out


def test_reference_yelp_app_builds_unmodified_and_runs_config1_plumbing(host_bins, tmp_path):
    yelp = os.path.join(host_bins, "yelp")
    if not os.path.exists(yelp):
        pytest.skip("reference sources not present on this box and no prebuilt build/yelp")
    write_ratings(tmp_path / "yelp_10core.txt")
    # every task except "test" ends in yelp.cpp:102-104's `else { return -1; }` after doing its work
    assert run([yelp, "--task=prepare"], tmp_path)[0] == 255
    assert os.path.exists(tmp_path / "yelp.bin")
    assert run([yelp, "--task=split"], tmp_path)[0] == 255
    assert os.path.exists(tmp_path / "yelp.train.bin") and os.path.exists(tmp_path / "yelp.test.bin")
    rc, out = run([yelp, "--task=test", "--method=NONE"], tmp_path)
    assert rc == 0, out
    rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)]
    assert len(rows) == 2                                  # iteration 0 and 1 of Solver<Popularity>
    recall10 = float(rows[-1].split("|")[8])
    assert 0.05 < recall10 < 0.6
    # the reference's --task=train falls through to `return -1` (yelp.cpp:88-104)
    assert run([yelp, "--task=train"], tmp_path)[0] != 0
    # a truncated cache must abort (glog CHECK convention), not train on zero-filled columns: user 0 / item 0 are valid ids
    blob = open(tmp_path / "yelp.bin", "rb").read()
    open(tmp_path / "yelp.bin", "wb").write(blob[:len(blob) - len(blob) // 3])
    rc, out = run([yelp, "--task=split"], tmp_path)
    assert rc not in (0, 255) and "truncated" in out, (rc, out[-500:])


REF_SAMPLE = "/root/reference/test/test_data/sample_movielens_data.txt"


def test_the_references_own_data_fixture_gives_what_its_data_test_asserts(host_bins, tmp_path):
    """The one data fixture the reference holds (test/test_data/sample_movielens_data.txt) through this repository's ingest, with the
    assertions of the reference's test/data_test.hpp:17-62: the "::" parser yields four fields per line (CHECK_EQ inside host_check),
    every instance has two features, 200 instances, the cache round-trips, random_split(0.3) leaves size * 0.7 and size * 0.3.
    The expected numbers are committed here; the file itself stays in the reference tree (this test runs where that tree exists).
    Beyond the reference's assertions: 5 users / 175 items / label sum 792 (counted from the file with plain Python below) and the
    per-user split's floor(0.3 n_u) (data-inl.hpp:249-261): 58 of the users' 22 + 20 + 33 + 38 + 87 ratings."""
    if not os.path.exists(REF_SAMPLE):
        pytest.skip("the reference tree is not on this box")
    rows = [l.strip().split("::") for l in open(REF_SAMPLE) if l.strip()]
    assert len(rows) == 200 and {len(r) for r in rows} == {4}
    local = tmp_path / "sample_movielens_data.txt"
    local.write_text(open(REF_SAMPLE).read())
    rc, out = run([os.path.join(host_bins, "host_check"), f"--movielens_sample={local}"], tmp_path)
    assert rc == 0, out
    m = re.search(r"movielens sample: instances (\d+) users (\d+) items (\d+) label_sum (\S+) by_user_split (\d+) (\d+) random_split (\d+) (\d+)", out)
    assert m, out
    inst, users, items, label_sum, utr, ute, rtr, rte = m.groups()
    assert int(inst) == 200                                             # data_test.hpp:52
    assert (int(rtr), int(rte)) == (140, 60)                            # data_test.hpp:58-59: size * 0.7, size * 0.3
    assert int(users) == len({r[0] for r in rows}) == 5 and int(items) == len({r[1] for r in rows}) == 175
    assert float(label_sum) == sum(float(r[2]) for r in rows) == 792.0
    per_user = np.unique([r[0] for r in rows], return_counts=True)[1]
    assert sorted(per_user.tolist()) == [20, 22, 33, 38, 87]
    assert int(ute) == int(np.floor(0.3 * per_user).sum()) == 58 and int(utr) == 142


def test_text_ingest_split_and_csr_are_pinned(host_bins, tmp_path):
    """SURVEY.md §8(f) rank 2, the data path in front of cdae_hip_set_interactions, against an independent numpy derivation:
      * text -> columns: ids in FIRST-SEEN order (instance-inl.hpp:22-37), one (user, item, 1) triple per line, file order
      * per-user split: floor(0.2 n) of every user's ratings to test, the rest to train, nothing lost or invented
        (data-inl.hpp:249-261)
      * Data::to_csr == sorted unique items per user of exactly the cached train / test ratings (data-inl.hpp:414-429)."""
    from helpers import read_data_cache, csr_of
    write_ratings(tmp_path / "ratings.txt")
    rc, out = run([os.path.join(host_bins, "host_check"), f"--input_file={tmp_path / 'ratings.txt'}", f"--dump_csr={tmp_path / 'csr'}"], tmp_path)
    assert rc == 0, out
    # ---- text -> columns
    users, items, uid, iid = [], [], {}, {}
    for line in open(tmp_path / "ratings.txt").read().splitlines()[1:]:
        u, i = line.split()
        users.append(uid.setdefault(u, len(uid)))
        items.append(iid.setdefault(i, len(iid)))
    whole = read_data_cache(tmp_path / "ratings.txt.bin")
    assert whole["user_names"] == list(uid) and whole["item_names"] == list(iid)
    np.testing.assert_array_equal(whole["users"], np.array(users, np.uint32))
    np.testing.assert_array_equal(whole["items"], np.array(items, np.uint32))
    assert (whole["labels"] == 1.0).all()
    # ---- split
    tr, te = read_data_cache(tmp_path / "csr.train.bin"), read_data_cache(tmp_path / "csr.test.bin")
    U, I = len(uid), len(iid)
    pair = lambda d: np.sort(d["users"].astype(np.int64) << 32 | d["items"].astype(np.int64))
    np.testing.assert_array_equal(np.sort(np.r_[pair(tr), pair(te)]), pair(whole))
    n_all, n_te = np.bincount(whole["users"], minlength=U), np.bincount(te["users"], minlength=U)
    np.testing.assert_array_equal(n_te, np.floor(0.2 * n_all).astype(np.int64))
    # ---- CSR
    for name, d in (("train", tr), ("test", te)):
        raw = open(tmp_path / f"csr.{name}", "rb").read()
        rows = int(np.frombuffer(raw, np.uint64, 1)[0])
        ptr = np.frombuffer(raw, np.int64, rows + 1, 8)
        col = np.frombuffer(raw, np.uint32, int(ptr[-1]), 8 + 8 * (rows + 1))
        eptr, ecol = csr_of(d["users"], d["items"], U)
        assert rows == U and col.max() < I
        np.testing.assert_array_equal(ptr, eptr)
        np.testing.assert_array_equal(col, ecol)


def _app_tables(yelp, tmp_path, flags, env):
    rc, out = run([yelp, "--task=test", "--method=CDAE"] + flags, tmp_path, env=env)
    assert rc == 0, out[-3000:]
    rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)][2:]           # after the two Popularity rows
    assert len(rows) == 51
    loss = np.array([float(r.split("|")[2]) for r in rows])
    topn = np.array([[float(x) for x in r.split("|")[3:11]] for r in rows])
    m = re.search(r"(\d+) interactions, csr fnv1a64 (\d+)", out)
    return loss, topn, int(m.group(1)), int(m.group(2))


@pytest.mark.gpu
@pytest.mark.parametrize("batch_users", [1, 64])
def test_reference_yelp_app_table_matches_the_oracle_on_the_same_rows(host_bins, tmp_path, batch_users):
    """Boundary-level parity (solver-inl.hpp:51-69): the UNMODIFIED yelp app's `Train Loss` and TOPN columns, iteration by
    iteration, against the CPU oracle run on the rows the app itself split (its train / test caches) with the same CDAE_SEED.
    batch_users = 1: the oracle's LITERAL schedule (cdae.hpp:136-358); 64: its block schedule.  The CSR handed to the device is
    pinned through the checksum the host class logs."""
    from helpers import read_data_cache, csr_of, fnv1a64
    import oracle as orc
    from oracle import binding as ob
    yelp = os.path.join(host_bins, "yelp")
    if not os.path.exists(yelp):
        pytest.skip("no build/yelp (reference sources were not present at build time)")
    write_ratings(tmp_path / "yelp_10core.txt")
    for task in ("prepare", "split"):
        assert run([yelp, f"--task={task}"], tmp_path)[0] == 255
    seed, K = 11, 50
    loss, topn, nnz, checksum = _app_tables(yelp, tmp_path, [f"--num_dim={K}", "--loss_type=CE", "--cratio=0.4", "--scaled=true", "--beta=1"],
                                            {"CDAE_SEED": str(seed), "CDAE_BATCH_USERS": str(batch_users)})
    tr, te = read_data_cache(tmp_path / "yelp.train.bin"), read_data_cache(tmp_path / "yelp.test.bin")
    U, I = len(tr["user_names"]), len(tr["item_names"])
    ptr, col = csr_of(tr["users"], tr["items"], U)
    tptr, tcol = csr_of(te["users"], te["items"], U)
    assert nnz == col.size and checksum == fnv1a64(ptr, col)              # the rows the device trained on are these rows
    o = orc.Oracle(orc.OracleConfig(num_dim=K, loss_type=ob.LOSS_CE, corruption_ratio=0.4, scaled=True, beta=1.0, learn_rate=0.1,
                                    lambda_=0.01, num_neg=5, num_corruptions=1), U, I, ptr, col)
    o.init_params(seed)
    exp_loss, exp_topn = [0.0], [orc.eval_topn(o.recommend(10), tptr, tcol)]
    for ep in range(50):
        if batch_users == 1:
            o.train_literal(seed, ep)
        else:
            o.train_batched(seed, ep, batch_users)
        exp_loss.append(o.data_loss(seed, ep + 1) + o.penalty_loss())    # the host class reports the loss with the NEXT epoch's masks
        exp_topn.append(orc.eval_topn(o.recommend(10), tptr, tcol))
    exp_loss, exp_topn = np.array(exp_loss), np.array(exp_topn)
    rel = np.abs(loss[1:] / exp_loss[1:] - 1)
    d_top = np.abs(topn - exp_topn)
    print(f"\nbatch_users {batch_users}: max rel loss diff, iterations 1-10 {rel[:10].max():.2e}, 1-50 {rel.max():.2e}; "
          f"max |d| of the TOPN columns {d_top.max(axis=0).round(5)}; rows with identical R@10: {(d_top[:, 5] < 6e-6).sum()}/51")
    assert loss[0] == 0.0
    assert rel[:10].max() <= 2e-4 and rel.max() <= 2e-3
    # one hit more or less on one of 300 users moves R@10 by ~0.0007 (fp32 device scores vs fp64 near ties); allow two
    assert d_top[0].max() <= 6e-6                                         # untrained model: same initial parameters, same lists
    assert d_top[:, 5].max() <= 0.0015 and d_top[:11, 5].max() <= 0.0008


@pytest.mark.gpu
def test_solver_cdae_trains_on_gpu_through_host_layer(host_bins, tmp_path):
    write_ratings(tmp_path / "ratings.txt")
    rc, out = run([os.path.join(host_bins, "host_check"), f"--input_file={tmp_path / 'ratings.txt'}", "--run_cdae=true",
                   "--num_dim=16", "--iters=6"], tmp_path, env={"CDAE_SEED": "7", "CDAE_BATCH_USERS": "64"})
    assert rc == 0, out
    rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)]
    cdae_rows = rows[2:]                                   # after the two Popularity rows
    assert len(cdae_rows) == 7
    losses = [float(r.split("|")[2]) for r in cdae_rows[1:]]
    assert all(np.isfinite(losses)) and min(losses) > 0    # (the positives-only loss is not monotone, cdae.hpp:93-96)
    assert float(cdae_rows[-1].split("|")[8]) > float(cdae_rows[0].split("|")[8])     # Recall@10 improves over untrained


@pytest.mark.gpu
def test_reference_yelp_app_trains_cdae_on_gpu(host_bins, tmp_path):
    yelp = os.path.join(host_bins, "yelp")
    if not os.path.exists(yelp):
        pytest.skip("no build/yelp (reference sources were not present at build time)")
    write_ratings(tmp_path / "yelp_10core.txt")
    for task in ("prepare", "split"):
        assert run([yelp, f"--task={task}"], tmp_path)[0] == 255
    rc, out = run([yelp, "--task=test", "--method=CDAE", "--num_dim=50", "--loss_type=CE", "--cratio=0.4", "--scaled=true",
                   "--beta=1"], tmp_path, env={"CDAE_SEED": "11", "CDAE_BATCH_USERS": "64"})
    assert rc == 0, out[-3000:]
    rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)]
    assert len(rows) == 2 + 51                             # Popularity (0,1) + CDAE iterations 0..50 (yelp.cpp:197)
    pop_r10 = float(rows[1].split("|")[8])
    best = max(float(r.split("|")[8]) for r in rows[2:])
    assert best > pop_r10, (best, pop_r10)


@pytest.mark.gpu
def test_reference_yelp_app_trains_cdae_with_linear_function_gate(host_bins, tmp_path):
    """--linear_function=true (the one CDAE flag cdae.sh never switches on): the gate Uu runs on the GPU too."""
    yelp = os.path.join(host_bins, "yelp")
    if not os.path.exists(yelp):
        pytest.skip("no build/yelp (reference sources were not present at build time)")
    write_ratings(tmp_path / "yelp_10core.txt")
    for task in ("prepare", "split"):
        assert run([yelp, f"--task={task}"], tmp_path)[0] == 255
    rc, out = run([yelp, "--task=test", "--method=CDAE", "--num_dim=20", "--loss_type=SQUARE", "--cratio=0.5", "--scaled=true",
                   "--beta=1", "--linear_function=true"], tmp_path, env={"CDAE_SEED": "11", "CDAE_BATCH_USERS": "32"})
    assert rc == 0, out[-3000:]
    assert "LinearFunction: 1" in out
    rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)]
    assert len(rows) == 2 + 51
    losses = [float(r.split("|")[2]) for r in rows[3:]]
    assert all(np.isfinite(losses))
    assert max(float(r.split("|")[8]) for r in rows[2:]) > float(rows[2].split("|")[8])      # improves over the untrained model


@pytest.mark.gpu
def test_reference_yelp_app_trains_cdae_full_output_k50(host_bins, tmp_path):
    """BASELINE configs[1] through the drop-in boundary: the unmodified yelp app, K=50 sigmoid + CE, with the full-output
    (bf16 MFMA) decode switched on by CDAE_FULL_OUTPUT=1 — every unrated item is a negative, --num_neg is ignored."""
    yelp = os.path.join(host_bins, "yelp")
    if not os.path.exists(yelp):
        pytest.skip("no build/yelp (reference sources were not present at build time)")
    write_ratings(tmp_path / "yelp_10core.txt")
    for task in ("prepare", "split"):
        assert run([yelp, f"--task={task}"], tmp_path)[0] == 255
    rc, out = run([yelp, "--task=test", "--method=CDAE", "--num_dim=50", "--loss_type=CE", "--cratio=0.5", "--scaled=true",
                   "--beta=1"], tmp_path, env={"CDAE_SEED": "11", "CDAE_BATCH_USERS": "64", "CDAE_FULL_OUTPUT": "1"})
    assert rc == 0, out[-3000:]
    rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)]
    assert len(rows) == 2 + 51
    losses = [float(r.split("|")[2]) for r in rows[3:]]
    assert all(np.isfinite(losses))
    pop_r10 = float(rows[1].split("|")[8])
    assert max(float(r.split("|")[8]) for r in rows[2:]) > pop_r10


@pytest.mark.gpu
def test_reference_yelp_app_trains_data_parallel_through_the_c_abi(host_bins, tmp_path):
    """CDAE_DEVICES=0,0 CDAE_LAYOUT=users: the UNMODIFIED yelp app drives Solver<CDAE>::train over two user shards through
    cdae_hip_multi_* (logical shards of GPU 0 here; distinct ids = one shard per GPU with a library-owned RCCL communicator),
    synchronous and pipelined exchange.  Loss and TOPN rows come from the sharded model; identical seeds give identical tables.
    The user-sharded delta exchange is taken only when asked for BY NAME (round 4: CDAE_DEVICES alone selects the item-rows layout)."""
    yelp = os.path.join(host_bins, "yelp")
    if not os.path.exists(yelp):
        pytest.skip("no build/yelp (reference sources were not present at build time)")
    write_ratings(tmp_path / "yelp_10core.txt")
    for task in ("prepare", "split"):
        assert run([yelp, f"--task={task}"], tmp_path)[0] == 255
    tables = []
    for env in ({"CDAE_DEVICES": "0,0", "CDAE_LAYOUT": "users"}, {"CDAE_DEVICES": "0,0", "CDAE_LAYOUT": "users"},
                {"CDAE_DEVICES": "0,0,0", "CDAE_EXCHANGE_EVERY": "2", "CDAE_LAYOUT": "users"}):
        rc, out = run([yelp, "--task=test", "--method=CDAE", "--num_dim=50", "--loss_type=CE", "--cratio=0.4", "--scaled=true",
                       "--beta=1"], tmp_path, env={"CDAE_SEED": "11", "CDAE_BATCH_USERS": "32", **env})
        assert rc == 0, out[-3000:]
        assert f"{len(env['CDAE_DEVICES'].split(','))} user shards" in out
        rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)]
        assert len(rows) == 2 + 51
        losses = [float(r.split("|")[2]) for r in rows[3:]]
        assert all(np.isfinite(losses))
        pop_r10 = float(rows[1].split("|")[8])
        assert max(float(r.split("|")[8]) for r in rows[2:]) > pop_r10
        tables.append([r.split("|")[2:10] for r in rows[2:]])      # loss + the eight TOPN columns (not the time columns)
    assert tables[0] == tables[1]                                  # deterministic: same shards, same seed, same table


@pytest.mark.gpu
def test_reference_yelp_app_trains_full_output_in_the_item_rows_layout(host_bins, tmp_path):
    """CDAE_LAYOUT=item_rows + CDAE_FULL_OUTPUT=1 + CDAE_DEVICES: the unmodified yelp app on the configs[4] layout (logical shards of
    GPU 0).  It is the single-GPU full-output schedule, so its table tracks the single-GPU run's."""
    yelp = os.path.join(host_bins, "yelp")
    if not os.path.exists(yelp):
        pytest.skip("no build/yelp (reference sources were not present at build time)")
    write_ratings(tmp_path / "yelp_10core.txt")
    for task in ("prepare", "split"):
        assert run([yelp, f"--task={task}"], tmp_path)[0] == 255
    tables = []
    for env in ({}, {"CDAE_DEVICES": "0,0,0", "CDAE_LAYOUT": "item_rows"}):
        rc, out = run([yelp, "--task=test", "--method=CDAE", "--num_dim=50", "--loss_type=CE", "--cratio=0.5", "--scaled=true",
                       "--beta=1"], tmp_path, env={"CDAE_SEED": "11", "CDAE_BATCH_USERS": "64", "CDAE_FULL_OUTPUT": "1", **env})
        assert rc == 0, out[-3000:]
        rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)]
        assert len(rows) == 2 + 51
        tables.append(np.array([[float(x) for x in r.split("|")[2:10]] for r in rows[2:]]))
    loss_a, loss_b = tables[0][1:, 0], tables[1][1:, 0]            # (iteration 0 is the untrained model: no loss printed)
    assert np.abs(loss_b / loss_a - 1).max() < 5e-3                # the same schedule: loss curves coincide
    assert np.abs(tables[1][:, 6] - tables[0][:, 6]).max() < 0.02  # Recall@10 column (300 users: one hit = 0.003)


@pytest.mark.gpu
def test_reference_yelp_app_trains_the_sampled_decode_in_the_item_rows_layout(host_bins, tmp_path):
    """CDAE_DEVICES alone (no CDAE_LAYOUT, no CDAE_FULL_OUTPUT): the unmodified yelp app takes the multi-GPU schedule that carries the
    single-GPU accuracy claim — the sampled decode over item shards (logical shards of GPU 0 here) — BY DEFAULT; CDAE_LAYOUT=item_rows
    names the same thing.  It is the single-GPU schedule, so its `Train Loss` / TOPN table is the single-GPU run's."""
    yelp = os.path.join(host_bins, "yelp")
    if not os.path.exists(yelp):
        pytest.skip("no build/yelp (reference sources were not present at build time)")
    write_ratings(tmp_path / "yelp_10core.txt")
    for task in ("prepare", "split"):
        assert run([yelp, f"--task={task}"], tmp_path)[0] == 255
    tables = []
    for env in ({}, {"CDAE_DEVICES": "0,0,0,0"}, {"CDAE_DEVICES": "0,0,0,0", "CDAE_LAYOUT": "item_rows"}):
        rc, out = run([yelp, "--task=test", "--method=CDAE", "--num_dim=50", "--loss_type=CE", "--cratio=0.5", "--scaled=true",
                       "--beta=1"], tmp_path, env={"CDAE_SEED": "11", "CDAE_BATCH_USERS": "64", **env})
        assert rc == 0, out[-3000:]
        assert ("item-row shards" in out) == bool(env)
        # round 6: the user is told that this layout does not speed the SAMPLED decode up (it is the exact schedule, not the fast one)
        assert ("SLOWER than one GPU" in out) == bool(env)
        rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)]
        assert len(rows) == 2 + 51
        tables.append(np.array([[float(x) for x in r.split("|")[2:10]] for r in rows[2:]]))
    assert np.array_equal(tables[1], tables[2])                    # the default IS the item-rows layout
    loss_a, loss_b = tables[0][1:, 0], tables[1][1:, 0]
    assert np.abs(loss_b / loss_a - 1).max() < 2e-4                # the same schedule up to fp32 association of two sums
    assert np.abs(tables[1][:, 6] - tables[0][:, 6]).max() < 0.003  # Recall@10 column (300 users: one hit on one user = 0.0007)


@pytest.mark.gpu
@pytest.mark.parametrize("method,loss", [("MF", "SQUARE"), ("MF", "CE"), ("BPR", "LOG"), ("BPR", "HINGE")])
def test_reference_yelp_app_trains_the_sibling_models_on_gpu(host_bins, tmp_path, method, loss):
    """--method=MF (libcf::IMF) and --method=BPR through the unmodified yelp app (yelp.cpp:122-165): both run on the GPU behind
    cdae_hip_create_mf.  Train Loss is 0 as in the reference (neither model overrides data_loss); Recall@10 must beat Popularity."""
    yelp = os.path.join(host_bins, "yelp")
    if not os.path.exists(yelp):
        pytest.skip("no build/yelp (reference sources were not present at build time)")
    write_ratings(tmp_path / "yelp_10core.txt")
    for task in ("prepare", "split"):
        assert run([yelp, f"--task={task}"], tmp_path)[0] == 255
    rc, out = run([yelp, "--task=test", f"--method={method}", "--num_dim=20", f"--loss_type={loss}"], tmp_path,
                  env={"CDAE_SEED": "11", "CDAE_BATCH_USERS": "16"})
    assert rc == 0, out[-3000:]
    assert ("BPR Model Configure" if method == "BPR" else "IMF Model Configure") in out
    rows = [l for l in out.splitlines() if re.search(r"\]\s+\d+\|", l)]
    assert len(rows) == 2 + 51
    pop_r10 = float(rows[1].split("|")[8])
    best = max(float(r.split("|")[8]) for r in rows[2:])
    assert best > pop_r10, (best, pop_r10)
