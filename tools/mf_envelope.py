#!/usr/bin/env python
"""Accuracy envelope of the IMF / BPR block schedule against the sequential loop (batch_users = 1, imf.hpp:71-115 / bpr.hpp:56-106).

    python tools/mf_envelope.py --shape yelp --num-dim 50 --batch-users 8 16 32 64 256 --seeds 20141119 7 1234 42 --epochs 10
Same data, init and sampling streams for every block size (the draws are keyed by user, not by block).  One JSON line per
(model, seed, batch_users): Recall@10 per epoch and users/s; then per (model, batch_users) the mean signed difference to the
sequential loop per epoch over the seeds and the largest single-seed |difference| — the form of the sampled CDAE path's claim
(DESIGN.md §2): "within +-0.002 as a mean over seeds".
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cdae_amd  # noqa: E402
from cdae_amd import synth  # noqa: E402


def curve(d, seed, K, lt, pairwise, B, epochs, num_neg=5):
    m = cdae_amd.MF(cdae_amd.MFConfig(num_dim=K, lt=lt, pairwise=pairwise, batch_users=B, num_neg=num_neg))
    m.reset(d, seed=seed)
    m.set_test_rows(d.test_ptr, d.test_col)
    rec, ups = [], []
    for ep in range(epochs):
        st = m.train_one_iteration(seed, ep)
        ups.append(st.users / st.wall_seconds)
        rec.append(float(m.eval_topn(10)[0][5]))
    m.close()
    return np.array(rec), float(np.median(ups))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="yelp")
    ap.add_argument("--num-dim", type=int, default=50)
    ap.add_argument("--batch-users", type=int, nargs="+", default=[8, 16, 32, 64, 256])
    ap.add_argument("--seeds", type=int, nargs="+", default=[20141119, 7, 1234, 42])
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--models", nargs="+", default=["IMF", "BPR"])
    ap.add_argument("--num-neg", type=int, default=5)
    args = ap.parse_args()
    kinds = {"IMF": (False, cdae_amd.SQUARE), "BPR": (True, cdae_amd.LOG)}
    for name in args.models:
        pairwise, lt = kinds[name]
        diffs = {B: [] for B in args.batch_users}
        speed = {B: [] for B in [1] + args.batch_users}
        for seed in args.seeds:
            d = synth.generate_shape(args.shape, seed=seed)
            lit, u1 = curve(d, seed, args.num_dim, lt, pairwise, 1, args.epochs, args.num_neg)
            speed[1].append(u1)
            print(json.dumps({"model": name, "shape": args.shape, "seed": seed, "batch_users": 1, "recall10": np.round(lit, 5).tolist(), "users_per_s": round(u1)}), flush=True)
            for B in args.batch_users:
                rec, ups = curve(d, seed, args.num_dim, lt, pairwise, B, args.epochs, args.num_neg)
                diffs[B].append(rec - lit)
                speed[B].append(ups)
                print(json.dumps({"model": name, "shape": args.shape, "seed": seed, "batch_users": B, "recall10": np.round(rec, 5).tolist(),
                                  "d_vs_sequential": np.round(rec - lit, 5).tolist(), "users_per_s": round(ups)}), flush=True)
        for B in args.batch_users:
            dd = np.array(diffs[B])
            print(json.dumps({"model": name, "shape": args.shape, "summary_batch_users": B, "seeds": len(args.seeds),
                              "mean_signed_d_per_epoch": np.round(dd.mean(axis=0), 5).tolist(), "max_abs_d_per_epoch": np.round(np.abs(dd).max(axis=0), 5).tolist(),
                              "users_per_s": round(float(np.median(speed[B]))), "sequential_users_per_s": round(float(np.median(speed[1])))}), flush=True)


if __name__ == "__main__":
    main()
