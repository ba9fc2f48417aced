#!/usr/bin/env python
"""How much does Recall@10 of the LITERAL schedule move with the random streams alone?

    python tools/reference_noise.py [--streams 8] [--epochs 5] [--batch-users 256 512] [--data-seed 20141119]

One data set (ML-10M shape, data seed 20141119); the HIP path at batch_users = 1 — the reference's own loop, which
tests/test_gpu_accuracy.py shows reproduces the fp64 literal oracle to six digits — is trained with different stream seeds
(initial values, dropout masks, negatives).  The spread of its Recall@10 is the resolution any "same accuracy as the
reference" statement has on this data set; the same stream seeds at the bench's batch_users give the paired differences.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cdae_amd  # noqa: E402
from cdae_amd import synth  # noqa: E402
import oracle as orc  # noqa: E402

HYPER = dict(num_neg=5, num_corruptions=1, corruption_ratio=0.5, scaled=True, learn_rate=0.1, beta=1.0, lambda_=0.01)


def curve(d, B, stream_seed, epochs):
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, batch_users=B, **HYPER))
    m.reset(d, seed=stream_seed)
    rec = []
    for ep in range(epochs):
        m.train_one_iteration(stream_seed, ep)
        rec.append(orc.eval_topn(m.recommend_all(10), d.test_ptr, d.test_col)[5])
    m.close()
    return np.array(rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--batch-users", type=int, nargs="+", default=[256])
    ap.add_argument("--data-seed", type=int, default=20141119)
    a = ap.parse_args()
    d = synth.generate_shape("ml10m", seed=a.data_seed)
    lit, bat = [], {B: [] for B in a.batch_users}
    for s in range(1, a.streams + 1):
        lit.append(curve(d, 1, 1000 + s, a.epochs))
        line = f"stream {1000 + s}: literal {np.round(lit[-1], 5)}"
        for B in a.batch_users:
            bat[B].append(curve(d, B, 1000 + s, a.epochs))
            line += f"  | {B}: d {np.round(bat[B][-1] - lit[-1], 5)}"
        print(line, flush=True)
    lit = np.array(lit)
    print(f"\nliteral schedule over {a.streams} stream seeds, per epoch:  mean {np.round(lit.mean(0), 5)}  std {np.round(lit.std(0, ddof=1), 5)}  "
          f"max-min {np.round(lit.max(0) - lit.min(0), 5)}")
    for B in a.batch_users:
        b = np.array(bat[B])
        dd = b - lit
        se = dd.std(0, ddof=1) / np.sqrt(a.streams)
        print(f"batch_users {B}: mean {np.round(b.mean(0), 5)}  std {np.round(b.std(0, ddof=1), 5)}")
        print(f"  paired difference (same streams):  mean {np.round(dd.mean(0), 5)}  std {np.round(dd.std(0, ddof=1), 5)}  max|d| {np.round(np.abs(dd).max(0), 5)}"
              f"  mean / standard error {np.round(dd.mean(0) / se, 2)}")


if __name__ == "__main__":
    main()
