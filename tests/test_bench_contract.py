"""-m gpu: bench.py's output contract (one JSON line, last on stdout, the keys the driver reads), on a small shape."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def run_bench(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--shape", "small", "--steps", "6", "--warmup", "2",
                          "--batch-users", "128", "--num-dim", "32"] + extra, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    return json.loads(lines[-1])          # the JSON line is the LAST line of stdout


def check(d, n_gpus=1):
    for k, t in REQUIRED.items():
        assert k in d, k
        assert isinstance(d[k], t) or (t is float and isinstance(d[k], int)), (k, d[k])
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["unit"] == "users/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["n_gpus"] == n_gpus and d["steps"] == 6 and d["warmup"] == 2 and d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9


def test_default_line_with_cpu_baseline(built):
    d = run_bench(["--cpu-users", "300"])
    check(d)
    assert d["dtype"] == "f32" and d["roofline"]["bound"] == "hbm"
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "users/s" and isinstance(c["sample"], str)


def test_full_output_line(built):
    d = run_bench(["--full-output", "--no-cpu-baseline"])
    check(d)
    assert d["dtype"] == "bf16" and d["roofline"]["bound"] == "mfma" and "cpu_baseline" not in d


def test_exchange_path_with_a_one_rank_rccl_group(built):
    """The N > 1 code path (process group after the handle, pipelined delta exchange, flush inside the timed region, JSON last
    after RCCL's banner) with a single rank — as much of `--gpus N` as one GPU can run."""
    d = run_bench(["--no-cpu-baseline"], env={"CDAE_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533",
                                              "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    check(d)
    assert d["config"]["exchange"] != "none"


def test_item_rows_layout_line(built):
    """--layout item-rows (the configs[4] layout) on one GPU with logical shards: same contract keys, strong scaling by definition"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--shape", "small", "--steps", "4", "--warmup", "1", "--batch-users", "128",
                          "--num-dim", "64", "--full-output", "--layout", "item-rows", "--logical-shards", "3", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    for k, t in REQUIRED.items():
        assert k in d, k
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["bound"] == "mfma"
    assert "item-rows x3" in d["config"]["parallelism"]


def test_certified_schedule_line_is_what_gpus_n_runs_by_default(built):
    """What `--gpus N` (N > 1) runs by default since round 6 — user shards on the relay + synchronous-exchange + global-accumulator
    schedule, one process driving every GPU, with the single-GPU and item-rows figures of the same node beside the line — exercised on
    one GPU as logical shards (the sum kernel in RCCL's place).  K timed steps are exactly K exchanged steps (cdae_hip_multi_train_steps)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--shape", "small", "--steps", "20", "--warmup", "3", "--batch-users", "128",
                          "--num-dim", "32", "--layout", "certified", "--logical-shards", "4", "--sync-batch-users", "32", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    for k, t in REQUIRED.items():
        assert k in d, k
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["steps"] == 20 and d["value"] > 0 and d["dtype"] == "f32"
    c = d["config"]
    assert c["batch_users"] == 32 and 120 <= c["global_batch"] <= 128 and c["schedule"]["relay_epochs"] == 1.0 and c["schedule"]["steps_per_epoch"] > 0   # (shards finish together: a step takes ceil(n_s / steps) of each)
    assert abs(d["value"] - c["global_batch"] * 1e3 / d["ms_per_step"]) < 0.02 * d["value"]
    side = c["same_node_alternatives"]
    assert side["single_gpu"]["users_per_s"] > 0 and side["item_rows_same_gpus"]["users_per_s"] > 0, side
    assert c["vs_single_gpu"] > 0 and "relay" in c["accuracy"]
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9


def test_sampled_item_rows_line(built):
    """The sampled decode in the item-rows layout — the multi-GPU schedule that is the single-GPU schedule exactly (the N > 1 default through
    round 5, now `--layout item-rows`) — exercised on one GPU as logical shards (--logical-shards selects the layout)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--shape", "small", "--steps", "20", "--warmup", "3", "--batch-users", "128",
                          "--num-dim", "32", "--logical-shards", "4", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    for k, t in REQUIRED.items():
        assert k in d, k
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["steps"] == 20 and d["value"] > 0 and d["dtype"] == "f32"
    assert d["roofline"]["bound"] == "hbm" and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    assert "item-rows x4" in d["config"]["parallelism"] and "single-GPU schedule exactly" in d["config"]["accuracy"]
