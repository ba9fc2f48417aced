#!/usr/bin/env python
"""Boundary-level timing: the UNMODIFIED apps/yelp/yelp.cpp (built in place against this repository's src/ headers) training CDAE on a
synthetic ML-10M-shape ratings file through libcdae_hip.so — what a user of the reference sees after the switch.

    python tools/app_bench.py [--shape ml10m] [--dir /tmp/app_bench] [--iters 4] [--num-dim 200] [--threads 8]

Writes the `user item` text file, runs --task=prepare / --task=split once, then --task=test --method=CDAE and reads the solver's own
table (solver-inl.hpp:24-69: `Iters | Time | Train Loss | TOPN…`, Time = cumulative wall seconds INCLUDING the loss pass and the
TOPN evaluation of every iteration).  Prints one JSON line: seconds per solver iteration, the users/s that corresponds to, and the
table.  Needs a GPU.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdae_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml10m")
    ap.add_argument("--dir", default="/tmp/app_bench")
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--num-dim", type=int, default=200)
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--seed", type=int, default=20141119)
    args = ap.parse_args()
    os.makedirs(args.dir, exist_ok=True)
    txt = os.path.join(args.dir, "yelp_10core.txt")
    d = synth.generate_shape(args.shape, seed=args.seed)
    users = np.r_[np.repeat(np.arange(d.num_users, dtype=np.uint32), np.diff(d.train_ptr)),
                  np.repeat(np.arange(d.num_users, dtype=np.uint32), np.diff(d.test_ptr))]
    items = np.r_[d.train_col, d.test_col]
    order = np.random.default_rng(1).permutation(users.size)
    import pandas as pd
    df = pd.DataFrame({"user": users[order], "item": items[order]})
    df["user"] = "u" + df["user"].astype(str)
    df["item"] = "i" + df["item"].astype(str)
    df.to_csv(txt, sep=" ", index=False)
    n_users, n_ratings = d.num_users, int(users.size)
    del df, users, items, order
    yelp = os.path.join(ROOT, "build", "yelp")
    out = {"shape": args.shape, "users": n_users, "ratings": n_ratings, "num_dim": args.num_dim, "threads": args.threads}
    for task in ("prepare", "split"):
        t0 = time.perf_counter()
        p = subprocess.run([yelp, f"--task={task}", f"--num_thread={args.threads}"], cwd=args.dir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode in (255, -1 & 0xFF), p.stdout[-2000:]
        out[f"{task}_s"] = round(time.perf_counter() - t0, 1)
    env = dict(os.environ)
    env.setdefault("CDAE_SEED", "7")
    # the app's CDAE solver always runs 50 iterations (yelp.cpp:197): read its table as it is printed and stop the process (this
    # child, by pid) once --iters rows are in
    t0 = time.perf_counter()
    p = subprocess.Popen([yelp, "--task=test", "--method=CDAE", f"--num_dim={args.num_dim}", "--loss_type=CE", "--cratio=0.5", "--scaled=true",
                          "--beta=1", f"--num_thread={args.threads}"], cwd=args.dir, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    rows, tail = [], []
    for line in p.stdout:
        tail.append(line)
        m = re.search(r"\]\s*(\d+)\|\s*([0-9.eE+-]+)\|\s*([0-9.eE+-]+)\|(.*)$", line)
        if m:
            if int(m.group(1)) == 0:
                rows = []                     # (the app prints the Popularity baseline's table first, yelp.cpp:108-113: keep the last table)
            rows.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), m.group(4).strip()))
            if len(rows) >= args.iters + 1 and rows[0][2] != rows[1][2]:
                p.kill()
                break
    p.wait()
    out["wall_s_until_last_row"] = round(time.perf_counter() - t0, 2)
    assert len(rows) >= 2, "".join(tail[-40:])
    times = [r[1] for r in rows]
    per_iter = np.diff(times)
    out["solver_rows"] = [{"iter": r[0], "time_s": r[1], "train_loss": r[2], "topn": r[3][:120]} for r in rows]
    out["seconds_per_solver_iteration"] = [round(float(x), 3) for x in per_iter]
    steady = float(np.median(per_iter[1:])) if per_iter.size > 1 else float(per_iter[0])
    out["steady_seconds_per_iteration"] = round(steady, 3)
    out["users_per_s_incl_loss_and_topn"] = round(n_users / steady)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
