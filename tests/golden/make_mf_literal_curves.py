#!/usr/bin/env python
"""Recall@10 curves of the SEQUENTIAL IMF / BPR loop (imf.hpp:71-115, bpr.hpp:56-106) — the fp64 oracle's literal restatement
(oracle/mf_oracle.cpp, train_literal) — at a BASELINE shape, one .npz per (model, seed): the anchor of the accuracy bound the IMF
library default block is held to (tests/test_gpu_mf.py).  One core, ~60 s per epoch at ML-10M shape K=200.

    python tests/golden/make_mf_literal_curves.py --model IMF --shape ml10m --num-dim 200 --seed 20141119 --epochs 5
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cdae_amd  # noqa: E402  (loss-type constants only)
from cdae_amd import synth  # noqa: E402
import oracle as orc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="IMF", choices=["IMF", "BPR"])
    ap.add_argument("--shape", default="ml10m")
    ap.add_argument("--num-dim", type=int, default=200)
    ap.add_argument("--seed", type=int, default=20141119)
    ap.add_argument("--epochs", type=int, default=5)
    args = ap.parse_args()
    pairwise = args.model == "BPR"
    lt = cdae_amd.LOG if pairwise else cdae_amd.SQUARE
    d = synth.generate_shape(args.shape, seed=args.seed)
    o = orc.MfOracle(orc.MfConfig(num_dim=args.num_dim, loss_type=lt, pairwise=pairwise), d.num_users, d.num_items, d.train_ptr, d.train_col)
    o.init_params(args.seed)
    rec, secs, topn = [], [], []
    for ep in range(args.epochs):
        t0 = time.perf_counter()
        o.train_literal(args.seed, ep)
        secs.append(time.perf_counter() - t0)
        cols = orc.eval_topn(o.recommend(10), d.test_ptr, d.test_col)
        topn.append(cols)
        rec.append(float(cols[5]))
        print(args.model, args.shape, args.seed, ep, rec[-1], round(secs[-1], 1), flush=True)
    out = os.path.join(ROOT, "tests", "golden", f"{args.shape}_k{args.num_dim}_{args.model.lower()}_seq_seed{args.seed}.npz")
    np.savez_compressed(out, model=args.model, shape=args.shape, seed=args.seed, num_dim=args.num_dim, recall10=np.array(rec), topn=np.array(topn),
                        train_seconds=np.array(secs), nnz_train=d.nnz_train)
    print("wrote", out)


if __name__ == "__main__":
    main()
