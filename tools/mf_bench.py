#!/usr/bin/env python
"""Throughput of the IMF / BPR paths at a BASELINE shape, next to the oracle's literal loop on one core.

    python tools/mf_bench.py --shape ml10m --num-dim 200 --batch-users 1 64 256 1024 4096 [--cpu-users 2000]
One JSON line per (model, batch_users): users/s of cdae_hip_train_epoch (one epoch after a warm-up epoch), Recall@10 after it.
batch_users = 1 is the reference's sequential loop; the library defaults are blocks of 16 (IMF) / 8 (BPR) users on BASELINE-sized data
sets (the certified block sizes); larger blocks are the activity-grouped block schedule as a throughput setting.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cdae_amd  # noqa: E402
from cdae_amd import synth  # noqa: E402
import oracle as orc  # noqa: E402  (metric function and the CPU baseline only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml10m")
    ap.add_argument("--num-dim", type=int, default=200)
    ap.add_argument("--batch-users", type=int, nargs="+", default=[1, 8, 16, 256, 1024, 4096])
    ap.add_argument("--cpu-users", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=20141119)
    args = ap.parse_args()
    d = synth.generate_shape(args.shape, seed=args.seed)
    for pairwise, lt, name in ((False, cdae_amd.SQUARE, "IMF"), (True, cdae_amd.LOG, "BPR")):
        for B in args.batch_users:
            m = cdae_amd.MF(cdae_amd.MFConfig(num_dim=args.num_dim, lt=lt, pairwise=pairwise, batch_users=B))
            m.reset(d, seed=args.seed)
            m.train_one_iteration(args.seed, 0)
            st = m.train_one_iteration(args.seed, 1)
            rec = float(orc.eval_topn(m.recommend_all(10), d.test_ptr, d.test_col)[5])
            print(json.dumps({"model": name, "batch_users": B, "library_default": B == int(cdae_amd.binding.load_library().cdae_hip_mf_default_batch_users(d.num_users, int(pairwise))), "users_per_s": round(st.users / st.wall_seconds), "ms_per_block": round(1e3 * st.wall_seconds / st.batches, 3),
                              "recall10_after_2_epochs": round(rec, 5)}), flush=True)
            m.close()
        n = min(args.cpu_users, d.num_users)
        o = orc.MfOracle(orc.MfConfig(num_dim=args.num_dim, loss_type=lt, pairwise=pairwise), d.num_users, d.num_items, d.train_ptr, d.train_col)
        o.init_params(args.seed)
        t0 = time.perf_counter()
        o.train_literal(args.seed, 0, 0, n)
        dt = time.perf_counter() - t0
        print(json.dumps({"model": name, "cpu_literal_users_per_s": round(n / dt, 1), "cores": 1, "sample_users": n}), flush=True)


if __name__ == "__main__":
    main()
