#!/usr/bin/env python
"""Recall@10 / loss-curve parity at benchmark shape: HIP path (several batch_users) vs the oracle's literal
(reference-semantics, fp64, sequential) schedule on identical data, parameters and random streams.

    python tools/recall_parity.py --shape ml10m --epochs 5 --batch-users 1024 4096 [--users N]

Prints one JSON line per run; used for BASELINE.md / DESIGN.md and the north-star +-0.002 Recall@10 check.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cdae_amd  # noqa: E402
from cdae_amd import synth  # noqa: E402
import oracle as orc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml10m")
    ap.add_argument("--users", type=int, default=0, help="truncate to the first N users (0 = all)")
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--num-dim", type=int, default=200)
    ap.add_argument("--batch-users", type=int, nargs="+", default=[1024, 4096])
    ap.add_argument("--loss", default="CE")
    ap.add_argument("--skip-oracle", action="store_true")
    ap.add_argument("--seed", type=int, default=20141119)
    args = ap.parse_args()

    d = synth.generate_shape(args.shape, seed=args.seed)
    if args.users:
        d = d.user_range(0, args.users)
    lt = cdae_amd.CROSS_ENTROPY if args.loss == "CE" else cdae_amd.SQUARE
    K = args.num_dim
    hyper = dict(num_neg=5, num_corruptions=1, corruption_ratio=0.5, scaled=True, learn_rate=0.1, beta=1.0, lambda_=0.01)

    base = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=lt, batch_users=1024, **hyper))
    base.reset(d, seed=args.seed)
    init = {w: base.get(w) for w in range(10) if w not in (2, 3)}
    base.close()

    def report(tag, recs, losses, secs):
        print(json.dumps({"run": tag, "recall10": [round(float(r), 5) for r in recs], "loss": [round(float(x), 1) for x in losses],
                          "train_seconds_per_epoch": round(secs, 3)}), flush=True)

    for B in args.batch_users:
        m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=lt, batch_users=B, **hyper))
        m.set_interactions(d.num_users, d.num_items, d.train_ptr, d.train_col)
        for w, v in init.items():
            m.set(w, v)
        recs, losses, t_train = [], [], 0.0
        for ep in range(args.epochs):
            st = m.train_one_iteration(args.seed, ep)
            t_train += st.wall_seconds
            losses.append(m.current_loss(args.seed, ep))
            recs.append(orc.eval_topn(m.recommend_all(10), d.test_ptr, d.test_col)[5])
        report(f"hip batch_users={B}", recs, losses, t_train / args.epochs)
        m.close()

    if not args.skip_oracle:
        o = orc.Oracle(orc.OracleConfig(num_dim=K, loss_type=lt, **hyper), d.num_users, d.num_items, d.train_ptr, d.train_col)
        o.init_params(args.seed)
        for w, v in init.items():
            o.set(w, v.astype(np.float64))
        recs, losses, t_train = [], [], 0.0
        for ep in range(args.epochs):
            t0 = time.time()
            o.train_literal(args.seed, ep)
            t_train += time.time() - t0
            losses.append(o.data_loss(args.seed, ep) + o.penalty_loss())
            recs.append(orc.eval_topn(o.recommend(10), d.test_ptr, d.test_col)[5])
            report(f"oracle literal (epochs so far {ep + 1})", recs, losses, t_train / (ep + 1))


if __name__ == "__main__":
    main()
