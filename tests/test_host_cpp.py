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
    """CDAE_DEVICES=0,0: the UNMODIFIED yelp app drives Solver<CDAE>::train over two user shards through cdae_hip_multi_*
    (logical shards of GPU 0 here; distinct ids = one shard per GPU with a library-owned RCCL communicator), synchronous
    and pipelined exchange.  Loss and TOPN rows come from the sharded model; identical seeds give identical tables."""
    yelp = os.path.join(host_bins, "yelp")
    if not os.path.exists(yelp):
        pytest.skip("no build/yelp (reference sources were not present at build time)")
    write_ratings(tmp_path / "yelp_10core.txt")
    for task in ("prepare", "split"):
        assert run([yelp, f"--task={task}"], tmp_path)[0] == 255
    tables = []
    for env in ({"CDAE_DEVICES": "0,0"}, {"CDAE_DEVICES": "0,0"}, {"CDAE_DEVICES": "0,0,0", "CDAE_EXCHANGE_EVERY": "2"}):
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
