"""CPU: bench.py's N > 1 watchdog — a stage that waits on another rank and never returns ends the process with ONE parseable JSON
line carrying "error" (exit status 4), not a hang; a single-GPU run arms nothing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_STALL = """
import sys, time, types
sys.path.insert(0, {root!r})
import bench
bench.WATCHDOG.configure(rank=1, world=int(sys.argv[1]), args=types.SimpleNamespace(steps=20, warmup=5))
with bench.WATCHDOG.stage("ncclCommInitRank", 0.3):
    time.sleep(float(sys.argv[2]))
print("returned")
"""


def run(world, sleep_s):
    return subprocess.run([sys.executable, "-c", _STALL.format(root=ROOT), str(world), str(sleep_s)], capture_output=True, text=True, timeout=60)


def test_a_stalled_stage_prints_an_error_line_and_exits():
    out = run(2, 5)
    assert out.returncode == 4, (out.returncode, out.stderr[-500:])
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    assert "ncclCommInitRank" in d["error"] and "rank 1" in d["error"] and d["value"] is None and d["n_gpus"] == 2 and d["steps"] == 20
    assert "returned" not in out.stdout


def test_a_stage_that_returns_in_time_is_left_alone():
    out = run(2, 0.01)
    assert out.returncode == 0 and out.stdout.strip().endswith("returned")


def test_one_gpu_arms_nothing():
    out = run(1, 0.6)
    assert out.returncode == 0 and out.stdout.strip().endswith("returned")
