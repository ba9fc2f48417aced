#!/usr/bin/env python
"""SURVEY.md §8(f) rank 2 at scale: time the data path in front of cdae_hip_set_interactions on a Netflix-shape ratings file.

    python tools/ingest_bench.py [--shape netflix] [--dir /tmp/ingest] [--threads 8]

Writes a `user item` text file of the named synthetic shape (shuffled line order, string ids — what apps/yelp reads,
yelp.cpp:60-66), then runs the UNMODIFIED yelp app's own tasks against this repository's host layer:
    --task=prepare   text -> first-seen id dictionaries + rating columns -> cache        (data-inl.hpp:13-80)
    --task=split     cache -> per-user 80/20 split -> train / test caches                (data-inl.hpp:231-272)
and build/host_check --csr_only, which loads the train cache and derives the CSR a model's reset() hands to the device
(Data::to_csr; recsys_model_base.hpp:29-34 builds a hashtable of hashtables here).  Reports wall seconds and peak RSS of each
step as one JSON line.  No GPU needed.
"""
import argparse
import json
import os
import resource
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdae_amd import synth  # noqa: E402


def timed(cmd, cwd, ok=(0,)):
    """wall seconds, peak RSS (GiB) of THIS child — /proc/<pid>/status VmHWM, polled: wait4's ru_maxrss also counts the pages the
    child shared with this Python process between fork and exec — and its combined output"""
    import tempfile
    import threading
    t0 = time.perf_counter()
    with tempfile.TemporaryFile(mode="w+") as logf:
        p = subprocess.Popen(cmd, cwd=cwd, stdout=logf, stderr=subprocess.STDOUT, text=True)
        peak = [0]

        def poll():
            while p.poll() is None:
                try:
                    for line in open(f"/proc/{p.pid}/status"):
                        if line.startswith("VmHWM:"):
                            peak[0] = max(peak[0], int(line.split()[1]))
                except OSError:
                    pass
                time.sleep(0.05)
        th = threading.Thread(target=poll)
        th.start()
        rc = p.wait()
        th.join()
        dt = time.perf_counter() - t0
        logf.seek(0)
        log = logf.read()
    rc = rc if rc >= 0 else 256 + rc
    assert (rc & 0xFF) in ok, log[-2000:]
    return dt, peak[0] / 2**20, log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="netflix")
    ap.add_argument("--dir", default="/tmp/ingest")
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--seed", type=int, default=20141119)
    ap.add_argument("--reuse-text", action="store_true", help="the ratings file of an earlier run is still in --dir")
    args = ap.parse_args()
    os.makedirs(args.dir, exist_ok=True)
    txt = os.path.join(args.dir, "yelp_10core.txt")
    t0 = time.perf_counter()
    if args.reuse_text and os.path.exists(txt):
        with open(txt, "rb") as f:
            n = sum(buf.count(b"\n") for buf in iter(lambda: f.read(1 << 24), b"")) - 1
    else:
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
        n = int(users.size)
        del df, users, items, order, d
    gen_s = time.perf_counter() - t0
    yelp, check = os.path.join(ROOT, "build", "yelp"), os.path.join(ROOT, "build", "host_check")
    out = {"shape": args.shape, "ratings": n, "text_bytes": os.path.getsize(txt), "generate_s": round(gen_s, 1), "threads": args.threads}
    dt, rss, _ = timed([yelp, "--task=prepare", f"--num_thread={args.threads}"], args.dir, ok=(255,))
    out["prepare_s"], out["prepare_peak_rss_gib"] = round(dt, 1), round(rss, 2)
    out["cache_bytes"] = os.path.getsize(os.path.join(args.dir, "yelp.bin"))
    dt, rss, _ = timed([yelp, "--task=split", f"--num_thread={args.threads}"], args.dir, ok=(255,))
    out["split_s"], out["split_peak_rss_gib"] = round(dt, 1), round(rss, 2)
    dt, rss, log = timed([check, f"--csr_only={os.path.join(args.dir, 'yelp.train.bin')}", f"--num_thread={args.threads}"], args.dir)
    out["load_train_cache_plus_to_csr_s"], out["csr_peak_rss_gib"] = round(dt, 1), round(rss, 2)
    for line in log.splitlines():
        if "csr_only:" in line:
            out["csr_only"] = line.split("csr_only:")[1].strip()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
