#!/usr/bin/env python
"""Accuracy fixtures at the BASELINE.json shapes, from the CPU oracle's LITERAL schedule (tests/golden/*_literal_*.npz).

    python tests/golden/make_literal_curves.py --shape ml10m --seed 20141119 --epochs 5

What the fixture pins: the table the reference prints per epoch — "Train Loss" = data_loss + penalty_loss
(/root/reference/src/solver/solver-inl.hpp:55) and the TOPN row, of which Recall@10 is rets[5]
(/root/reference/src/model/evaluation.hpp:183-219) — when `train_one_iteration` (cdae.hpp:136-146) runs strictly
user by user in fp64, on the synthetic data set `cdae_amd.synth.generate_shape(shape, seed)` with the counter-based
random streams of include/cdae_rng.h keyed by the same seed.  `tests/test_gpu_accuracy.py` trains the HIP path at
bench.py's default `batch_users` on the same data / init / streams and asserts |dRecall@10| <= 0.002 per epoch
(the north star's tolerance) and the loss-curve tolerance stated there.

The reference itself cannot be built in this image (Eigen / Boost / glog / gflags absent), so this is the oracle's
restatement, not the reference binary: "parity unpinned"  applies to these files too.

One ML-10M-shape epoch costs ~90 s of one core for training plus ~100 s for the fp64 top-10 of 70 000 users.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cdae_amd import synth  # noqa: E402
import oracle as orc  # noqa: E402
from oracle import binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# benchmark hyper-parameters — tests/test_gpu_accuracy.py and bench.py use the same
HYPER = dict(num_neg=5, num_corruptions=1, corruption_ratio=0.5, scaled=True, learn_rate=0.1, beta=1.0, lambda_=0.01)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml10m")
    ap.add_argument("--seed", type=int, default=20141119)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--num-dim", type=int, default=200)
    ap.add_argument("--loss", default="CE", choices=["CE", "SQUARE"])
    ap.add_argument("--full-output-batch", type=int, default=0,
                    help="> 0: the full-output block schedule (Oracle.train_full) with this many users per block instead of the literal one")
    ap.add_argument("--full-output-literal", action="store_true",
                    help="the B = 1 limit of the full-output decode — the reference loop cdae.hpp:225-293 fed EVERY unrated item as a negative "
                         "(= --full-output-batch 1; file tag `full1`): what tools/accuracy_envelope.py --full-output and "
                         "tests/test_gpu_accuracy.py measure the block schedule against")
    ap.add_argument("--data-seed", type=int, default=None,
                    help="seed of the synthetic DATA SET when it should differ from --seed (which then only keys the random streams: initial "
                         "values, dropout masks, negatives) — several stream seeds on one data set cost the GPU test one data generation")
    ap.add_argument("--eval-users", type=int, default=0,
                    help="> 0: Recall@10 over the first N users only (the fp64 top-10 of 480 000 x 17 700 x 200 is ~20 min of one core "
                         "per epoch; training and the reported loss always cover every user)")
    ap.add_argument("--tag-suffix", default="",
                    help="appended to the schedule tag of the file name (`literal50` = the app's own horizon, Solver<CDAE>(model, 50), "
                         "/root/reference/apps/yelp/yelp.cpp:197: kept apart from the 5-epoch six-seed set the default accuracy test globs)")
    ap.add_argument("--out-dir", default="",
                    help="write the fixture here instead of tests/golden (a long run saves after every epoch: a file that is still short of "
                         "what a test expects does not belong where the tests glob — copy it over when it is complete)")
    ap.add_argument("--checkpoint", default="",
                    help="directory for a resumable run: after every epoch the oracle's parameters and the curves so far are written there "
                         "(fp64, ~1.5 GB at Netflix shape: not a fixture, keep it out of the repository) and a later run with the same "
                         "arguments continues behind the last finished epoch — a 20-epoch Netflix-shape run is ~4.5 h of one core")
    args = ap.parse_args()
    if args.full_output_literal:
        args.full_output_batch = 1

    data_seed = args.seed if args.data_seed is None else args.data_seed
    d = synth.generate_shape(args.shape, seed=data_seed)
    lt = ob.LOSS_CE if args.loss == "CE" else ob.LOSS_SQUARE
    o = orc.Oracle(orc.OracleConfig(num_dim=args.num_dim, loss_type=lt, **HYPER), d.num_users, d.num_items, d.train_ptr, d.train_col)
    o.init_params(args.seed)
    rec10, loss, data_loss, metrics, secs = [], [], [], [], []
    tag = (f"full{args.full_output_batch}" if args.full_output_batch else "literal") + args.tag_suffix
    name = f"{args.shape}_k{args.num_dim}_{args.loss.lower()}_{tag}_seed{args.seed}" + ("" if args.data_seed is None else f"_data{data_seed}") + ".npz"

    def save(ne):
        # parameter probes (full-output fixtures compare parameters too: Recall is uninformative on a few hundred users)
        rng = np.random.default_rng(args.seed)
        pop = np.bincount(d.train_col, minlength=d.num_items)
        probe_items = np.unique(np.concatenate([np.argsort(-pop)[:32], rng.choice(d.num_items, 32, replace=False)])).astype(np.int64)
        probe_users = np.sort(rng.choice(d.num_users, min(16, d.num_users), replace=False)).astype(np.int64)
        K = args.num_dim
        probes = dict(probe_items=probe_items, probe_users=probe_users,
                      W_rows=o.get(ob.P_W).reshape(d.num_items, K)[probe_items], bp_rows=o.get(ob.P_BP)[probe_items],
                      Wu_rows=o.get(ob.P_WU).reshape(d.num_users, K)[probe_users], b=o.get(ob.P_B),
                      W_absmax=np.abs(o.get(ob.P_W)).max(), Wu_absmax=np.abs(o.get(ob.P_WU)).max(), bp_absmax=np.abs(o.get(ob.P_BP)).max())
        out_dir = args.out_dir or OUT
        os.makedirs(out_dir, exist_ok=True)
        tmp = os.path.join(out_dir, name + ".tmp.npz")
        np.savez(tmp, shape=args.shape, seed=args.seed, data_seed=data_seed, num_dim=args.num_dim, loss=args.loss,
                 full_output_batch=args.full_output_batch, hyper=np.array(sorted(HYPER.items()), dtype=object).astype(str),
                 recall10=np.array(rec10), train_loss=np.array(loss), data_loss=np.array(data_loss), topn=np.array(metrics),
                 train_seconds=np.array(secs), nnz_train=d.nnz_train, eval_users=ne, **probes)
        os.replace(tmp, os.path.join(out_dir, name))  # a long run (50 epochs = hours of one core) leaves a usable prefix if it is cut short

    first = 0
    ck = os.path.join(args.checkpoint, name[:-4]) if args.checkpoint else ""
    if ck and os.path.exists(os.path.join(ck, "curves.npz")):
        c = np.load(os.path.join(ck, "curves.npz"))
        rec10, loss, data_loss, secs = list(c["recall10"]), list(c["train_loss"]), list(c["data_loss"]), list(c["train_seconds"])
        metrics = [np.asarray(m) for m in c["topn"]]
        for w in range(ob.P_COUNT):
            f = os.path.join(ck, f"p{w}.npy")
            if os.path.exists(f):
                o.set(w, np.load(f))
        first = len(rec10)
        print(f"resuming behind epoch {first} from {ck}", flush=True)

    def checkpoint():
        if not ck:
            return
        os.makedirs(ck, exist_ok=True)
        for w in range(ob.P_COUNT):
            try:
                a = o.get(w)
            except Exception:
                continue
            if a is not None and a.size:
                np.save(os.path.join(ck, f"p{w}.tmp.npy"), a)
                os.replace(os.path.join(ck, f"p{w}.tmp.npy"), os.path.join(ck, f"p{w}.npy"))
        np.savez(os.path.join(ck, "curves.tmp.npz"), recall10=np.array(rec10), train_loss=np.array(loss), data_loss=np.array(data_loss),
                 topn=np.array(metrics), train_seconds=np.array(secs))
        os.replace(os.path.join(ck, "curves.tmp.npz"), os.path.join(ck, "curves.npz"))     # written last: the parameters above belong to it

    for ep in range(first, args.epochs):
        t0 = time.time()
        if args.full_output_batch:
            o.train_full(args.seed, ep, args.full_output_batch)
        else:
            o.train_literal(args.seed, ep)
        secs.append(time.time() - t0)
        dl = o.data_loss(args.seed, ep)
        data_loss.append(dl)
        loss.append(dl + o.penalty_loss())
        ne = min(args.eval_users, d.num_users) if args.eval_users else d.num_users
        m = orc.eval_topn(o.recommend(10, 0, ne), d.test_ptr[:ne + 1], d.test_col[:d.test_ptr[ne]])
        metrics.append(m)
        rec10.append(m[5])
        save(ne)
        checkpoint()
        print(f"[{args.shape} seed {args.seed}] epoch {ep + 1}: loss {loss[-1]:.1f} recall@10 {rec10[-1]:.5f} ({secs[-1]:.0f} s train)", flush=True)
    print("wrote", name)


if __name__ == "__main__":
    main()
