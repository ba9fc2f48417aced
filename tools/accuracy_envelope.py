#!/usr/bin/env python
"""Accuracy envelope of the HIP path against the committed literal-schedule fixtures (tests/golden/*_literal_seed*.npz).

    python tools/accuracy_envelope.py --batch-users 128 256 384 512 [--seeds 7 1234] [--shards 8 --period 2]

Single handle (default): trains the HIP path at each `batch_users` on the fixture's data / init / random streams and
prints, per epoch, Recall@10 and the reported train loss next to the fixture's, plus users/s of the training calls.
--shards N > 1: the data-parallel schedule on ONE GPU — the data set is split into N user shards (balanced by nnz like
cdae_amd.distributed.shard_bounds), N logical ranks train `batch_users` users each per step from replicated shared
parameters and exchange their accumulated deltas (cdae_hip_delta_stage / _merge on every shard, a sum over the staged buffers in between: synchronous when
--period 0, one period late otherwise).  One JSON line per run; DESIGN.md §2 / §7 tables are made from these lines.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cdae_amd  # noqa: E402
from cdae_amd import synth  # noqa: E402
import oracle as orc  # noqa: E402  (test infrastructure: only its metric function eval_topn is used here)

HYPER = dict(num_neg=5, num_corruptions=1, corruption_ratio=0.5, scaled=True, learn_rate=0.1, beta=1.0, lambda_=0.01)


def fixtures(shape, K, loss, seeds, tag="literal"):
    out = []
    for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", f"{shape}_k{K}_{loss.lower()}_{tag}_seed*.npz"))):
        f = np.load(p, allow_pickle=True)
        if seeds and int(f["seed"]) not in seeds:
            continue
        out.append(f)
    return out


def run_single(d, seed, K, lt, B, epochs, full_output=False, ne=None):
    m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=lt, batch_users=B, full_output=full_output, **HYPER))
    m.reset(d, seed=seed)
    rec, loss, secs = [], [], 0.0
    ne = d.num_users if ne is None else ne
    for ep in range(epochs):
        st = m.train_one_iteration(seed, ep)
        secs += st.wall_seconds
        loss.append(m.current_loss(seed, ep))
        rec.append(recall10(m, d, ne))
    m.close()
    return rec, loss, d.num_users * epochs / secs


def shard_cuts(row_ptr, num_users, shards):
    from cdae_amd.distributed import shard_bounds
    return [shard_bounds(num_users, shards, r, row_ptr) for r in range(shards)]


def recall10(m, d, ne):
    """Recall@10 over the first `ne` users (Netflix-shape fixtures score 60 000 of the 480 000)"""
    return float(orc.eval_topn(m.recommend_all(10, 0, ne), d.test_ptr[:ne + 1], d.test_col[:d.test_ptr[ne]])[5])


def run_schedule(d, seed, K, lt, B, epochs, shards, period, relay_epochs, combine, ne, handle_batch=256):
    """cdae_hip_multi_set_schedule on `shards` logical shards of GPU 0: `relay_epochs` epochs (fractions allowed) on the single-GPU
    schedule (blocks of `handle_batch` users, handed from shard to shard), then exchanged steps of B users per shard, `combine` rule."""
    m = cdae_amd.MultiCDAE(cdae_amd.CDAEConfig(num_dim=K, lt=lt, batch_users=handle_batch, **HYPER), devices=[0] * shards)
    m.set_schedule(period=period, combine=combine, sync_batch_users=B, relay_epochs=relay_epochs)
    m.reset(d, seed=seed)
    rec, loss, secs = [], [], 0.0
    for ep in range(epochs):
        secs += m.train_one_iteration(seed, ep).wall_seconds
        loss.append(m.current_loss(seed, ep))
        rec.append(recall10(m, d, ne))
    m.close()
    return rec, loss, d.num_users * epochs / secs


def run_multi(d, seed, K, lt, B, epochs, shards, period, warm_epochs=0, warm_batch=256):
    """the product path: cdae_hip_multi_* with `shards` logical shards of GPU 0 (same schedule as one shard per GPU; the
    all-reduce is the library's fixed-order sum kernel instead of RCCL).  warm_epochs > 0: the first epochs run the
    single-GPU schedule (batch_users = warm_batch) and its parameters are handed to the shards."""
    m = cdae_amd.MultiCDAE(cdae_amd.CDAEConfig(num_dim=K, lt=lt, batch_users=B, **HYPER), devices=[0] * shards, exchange_every=period)
    m.reset(d, seed=seed)
    rec, loss, secs = [], [], 0.0
    if warm_epochs:
        one = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=lt, batch_users=warm_batch, **HYPER))
        one.reset(d, seed=seed)
        for ep in range(warm_epochs):
            one.train_one_iteration(seed, ep)
            loss.append(one.current_loss(seed, ep))
            rec.append(float(orc.eval_topn(one.recommend_all(10), d.test_ptr, d.test_col)[5]))
        for w in (cdae_amd.P_W, cdae_amd.P_W_AG, cdae_amd.P_B, cdae_amd.P_B_AG, cdae_amd.P_BP, cdae_amd.P_BP_AG, cdae_amd.P_WU, cdae_amd.P_WU_AG):
            m.set(w, one.get(w))
        one.close()
    for ep in range(warm_epochs, epochs):
        secs += m.train_one_iteration(seed, ep).wall_seconds
        loss.append(m.current_loss(seed, ep))
        rec.append(float(orc.eval_topn(m.recommend_all(10), d.test_ptr, d.test_col)[5]))
    m.close()
    return rec, loss, d.num_users * epochs / secs


def run_sharded(d, seed, K, lt, B, epochs, shards, period, rule=0):
    """(touch-mean experiments only; the sum rule runs through run_multi) `shards` single-GPU handles on cuda:0 driven through the same C-ABI calls (cdae_hip_delta_stage / _merge) as the
    data-parallel ranks; the all-reduce(sum) between them is a torch sum over the staged buffers.  period 0 = synchronous."""
    import torch
    from cdae_amd.distributed import _DeviceBuffer
    dev = torch.device("cuda", 0)
    cuts = shard_cuts(d.train_ptr, d.num_users, shards)
    ms, sends, recvs = [], [], []
    for r, (u0, u1) in enumerate(cuts):
        sd = d.user_range(u0, u1)
        m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=lt, batch_users=B, **HYPER))
        m.set_interactions(sd.num_users, sd.num_items, sd.train_ptr, sd.train_col, user_id_offset=u0)
        m.init_params(seed)                      # shared block identical on every shard; Wu rows keyed by GLOBAL user id
        m.delta_begin(); m.delta_stage(); m.synchronize()
        ps, nfull = m.delta_device_ptr()
        pr, n = m.delta_recv_device_ptr()
        if rule:                                 # touch-mean: the synchronous API's [delta | touch] buffer, reduced in place
            sends.append(torch.as_tensor(_DeviceBuffer(ps, nfull), device=dev))
        else:
            sends.append(torch.as_tensor(_DeviceBuffer(ps, n), device=dev))
            recvs.append(torch.as_tensor(_DeviceBuffer(pr, n), device=dev))
        ms.append(m)
    sizes = [u1 - u0 for u0, u1 in cuts]
    steps = -(-max(sizes) // B)
    per = [-(-n // steps) for n in sizes]        # users per step of every shard: all shards finish an epoch together

    def reduce_all():
        for m in ms:
            m.synchronize()
        total = sends[0].clone()
        for t in sends[1:]:
            total += t
        for t in recvs:
            t.copy_(total)
        torch.cuda.synchronize()

    def boundary(pending, start_next):
        for m in ms:
            if pending and start_next:
                m.delta_merge_stage()
            elif pending:
                m.delta_merge()
            elif start_next:
                m.delta_stage()
        if start_next:
            reduce_all()

    rec, loss, secs = [], [], 0.0
    for ep in range(epochs):
        t0 = time.perf_counter()
        pending, batches = False, 0
        for t in range(steps):
            for r, m in enumerate(ms):
                a, b = min(sizes[r], t * per[r]), min(sizes[r], (t + 1) * per[r])
                if b > a:
                    m.enqueue_users(seed, ep, a, b)
            batches += 1
            if rule:                              # synchronous, CDAE_DELTA_TOUCH_MEAN (rows / #ranks that touched them)
                for m in ms:
                    m.delta_compute(); m.synchronize()
                total = sends[0].clone()
                for t in sends[1:]:
                    total += t
                for t in sends:
                    t.copy_(total)
                torch.cuda.synchronize()
                for m in ms:
                    m.delta_apply(shards, rule); m.delta_begin()
            elif period == 0:
                boundary(False, True); boundary(True, False)
            elif batches % period == 0:
                boundary(pending, True); pending = True
        if period and not rule:                   # flush: every shard ends the epoch with the same shared parameters
            boundary(pending, True); boundary(True, False)
        for m in ms:
            m.synchronize()
        secs += time.perf_counter() - t0
        # evaluation: the reported loss and the top-10 of every user, each from the shard that owns the user
        dl = sum(m.data_loss(seed, ep) for m in ms)
        pen = ms[0].penalty_loss() + sum(0.5 * HYPER["lambda_"] * float((m.get(cdae_amd.P_WU).astype(np.float64) ** 2).sum()) for m in ms[1:])
        loss.append(dl + pen)
        top = np.concatenate([m.recommend_all(10) for m in ms])
        rec.append(float(orc.eval_topn(top, d.test_ptr, d.test_col)[5]))
    for m in ms:
        m.close()
    return rec, loss, d.num_users * epochs / secs


def run_hybrid(d, seed, K, lt, B, epochs, shards, hot):
    """The hot-row / tail HYBRID of the data-parallel schedule, emulated at parameter level on one GPU (would
    owner-computed popular rows bring user shards + delta all-reduce inside the single-GPU accuracy envelope?).
    Per global step (every shard trains B of its users from the same replicated parameters):
      (b) the user-sharded step as the product runs it — `shards` handles, cdae_hip_delta_stage / sum / _merge, synchronous;
      (a) the SAME users through ONE handle that holds every user, shard range after shard range (batch_users = B): every item row
          takes its examples strictly one after the other — what the owner of a row would compute over the global batch;
      composed state = (b) for the tail rows of W / W_ag / b' / b'_ag, (a) for the `hot` most popular item rows, for the hidden bias
      b / b_ag (the other strictly sequential chain) and for the user node; both sides continue from the composed state.
    Optimistic by construction where it departs from a real implementation (the owner's chain sees z refreshed after every shard
    range instead of once per global step; Wu comes from the exact side), so a FAILING envelope here closes the question."""
    cuts = shard_cuts(d.train_ptr, d.num_users, shards)
    ms = []
    for r, (u0, u1) in enumerate(cuts):
        sd = d.user_range(u0, u1)
        m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=lt, batch_users=B, **HYPER))
        m.set_interactions(sd.num_users, sd.num_items, sd.train_ptr, sd.train_col, user_id_offset=u0)
        m.init_params(seed)
        ms.append(m)
    a = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=K, lt=lt, batch_users=B, **HYPER))
    a.reset(d, seed=seed)
    pop = np.bincount(d.train_col, minlength=d.num_items)
    H = np.argsort(-pop, kind="stable")[:hot]
    SHARED = (cdae_amd.P_W, cdae_amd.P_W_AG, cdae_amd.P_BP, cdae_amd.P_BP_AG)
    sizes = [u1 - u0 for u0, u1 in cuts]
    steps = -(-max(sizes) // B)
    per = [-(-n // steps) for n in sizes]
    rec, loss, secs = [], [], 0.0
    for ep in range(epochs):
        t0 = time.perf_counter()
        for t in range(steps):
            base = {w: ms[0].get(w) for w in SHARED + (cdae_amd.P_B, cdae_amd.P_B_AG)}          # replicas agree at a step's start
            # (b): every shard's users from the common state; the summed deltas of the shared block
            delta = {w: np.zeros_like(base[w], dtype=np.float64) for w in SHARED}
            for r, m in enumerate(ms):
                lo, hi = min(sizes[r], t * per[r]), min(sizes[r], (t + 1) * per[r])
                if hi > lo:
                    m.train_users(seed, ep, lo, hi)
                    for w in SHARED:
                        delta[w] += m.get(w).astype(np.float64) - base[w]
            # (a): the same users, shard range after shard range, through the handle that holds everybody
            for r, (u0, u1) in enumerate(cuts):
                lo, hi = min(sizes[r], t * per[r]), min(sizes[r], (t + 1) * per[r])
                if hi > lo:
                    a.train_users(seed, ep, u0 + lo, u0 + hi)
            comp = {w: (base[w].astype(np.float64) + delta[w]).astype(np.float32) for w in SHARED}
            for w in SHARED:
                exact = a.get(w)
                comp[w][H] = exact[H]
            comp[cdae_amd.P_B], comp[cdae_amd.P_B_AG] = a.get(cdae_amd.P_B), a.get(cdae_amd.P_B_AG)
            wu, wu_ag = a.get(cdae_amd.P_WU), a.get(cdae_amd.P_WU_AG)
            for w, v in comp.items():
                a.set(w, v)
            for r, (m, (u0, u1)) in enumerate(zip(ms, cuts)):
                for w, v in comp.items():
                    m.set(w, v)
                m.set(cdae_amd.P_WU, wu[u0:u1]); m.set(cdae_amd.P_WU_AG, wu_ag[u0:u1])
        secs += time.perf_counter() - t0
        loss.append(a.current_loss(seed, ep))
        rec.append(float(orc.eval_topn(a.recommend_all(10), d.test_ptr, d.test_col)[5]))
    for m in ms + [a]:
        m.close()
    return rec, loss, d.num_users * epochs / secs


def full_output_envelope(args, lt):
    """--full-output: the block schedule of the full-output decode (one summed step per decoder row per block of B users, DESIGN.md
    §5b) against its B = 1 limit — the reference loop cdae.hpp:225-293 fed every unrated item — per seed: Recall@10 / reported loss
    per epoch at every block size, the literal curve (committed fp64 fixture `*_full1_seed*.npz` when there is one, else the HIP
    path at batch_users = 1), the first epoch at which a block size reaches the literal's best Recall@10, and users/s."""
    for seed in (args.seeds or [20141119]):
        d = synth.generate_shape(args.shape, seed=seed)
        fx = os.path.join(ROOT, "tests", "golden", f"{args.shape}_k{args.num_dim}_{args.loss.lower()}_full1_seed{seed}.npz")
        ep = args.epochs or 20
        prev = None
        if args.literal_from:                       # the device's batch_users = 1 curve of an earlier envelope file (the B = 1 loop does not change
            for line in open(args.literal_from):    # when the block schedule does; 70 000 block steps per epoch at ML-10M shape are 3 min of GPU per seed)
                r = json.loads(line)
                if r.get("run") == "full-output literal" and r.get("shape") == args.shape and int(r.get("seed")) == seed:
                    prev = r
        if os.path.exists(fx) and len(np.load(fx, allow_pickle=True)["recall10"]) >= (args.literal_epochs or 0):
            f = np.load(fx, allow_pickle=True)
            lit_r, lit_l, src = [float(x) for x in f["recall10"]], [float(x) for x in f["train_loss"]], "fp64 fixture"
        elif prev is not None:
            lit_r, lit_l, src = prev["recall10"], prev["loss"], prev["source"] + " (from " + os.path.basename(args.literal_from) + ")"
        else:
            lit_r, lit_l, _ = run_single(d, seed, args.num_dim, lt, 1, args.literal_epochs or ep, full_output=True)
            src = "hip batch_users=1"
        best = max(lit_r)
        print(json.dumps({"run": "full-output literal", "source": src, "shape": args.shape, "seed": seed, "recall10": [round(x, 5) for x in lit_r],
                          "loss": [round(x, 1) for x in lit_l], "best_recall10": round(best, 5)}), flush=True)
        for B in args.batch_users:
            rec, loss, ups = run_single(d, seed, args.num_dim, lt, B, ep, full_output=True)
            n = min(len(rec), len(lit_r))
            reach = next((i + 1 for i, r in enumerate(rec) if r >= best), None)
            print(json.dumps({"run": "hip full-output", "shape": args.shape, "seed": seed, "batch_users": B, "recall10": [round(x, 5) for x in rec],
                              "loss": [round(x, 1) for x in loss], "d_recall_vs_literal_same_epoch": [round(rec[i] - lit_r[i], 5) for i in range(n)],
                              "epochs_to_literal_best": reach, "users_per_s": round(ups),
                              "seconds_to_literal_best": None if reach is None else round(reach * d.num_users / ups, 4)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml10m")
    ap.add_argument("--num-dim", type=int, default=200)
    ap.add_argument("--loss", default="CE")
    ap.add_argument("--batch-users", type=int, nargs="+", default=[512])
    ap.add_argument("--seeds", type=int, nargs="*", default=[])
    ap.add_argument("--epochs", type=int, default=0, help="0 = as many as the fixture holds")
    ap.add_argument("--shards", type=int, nargs="+", default=[1])
    ap.add_argument("--period", type=int, nargs="+", default=[0], help="exchange period of the sharded runs (0 = synchronous)")
    ap.add_argument("--rule", type=int, default=0, help="0 sum, 1 touch-mean (synchronous only)")
    ap.add_argument("--warm-epochs", type=int, default=0, help="sharded runs: first N epochs on the single-GPU schedule (batch_users 256)")
    ap.add_argument("--relay-epochs", type=float, nargs="+", default=[],
                    help="--shards N: cdae_hip_multi_set_schedule — this many epochs (fractions allowed) on the single-GPU schedule relayed "
                         "from shard to shard, then exchanged steps of --batch-users per shard (replaces --warm-epochs, which emulates it)")
    ap.add_argument("--combine", type=int, nargs="+", default=[0], help="with --relay-epochs: 0 sum, 1 global accumulator (CDAE_COMBINE_*)")
    ap.add_argument("--fixture-tag", default="literal", help="schedule tag of the fixture files (literal | literal50 | literal20)")
    ap.add_argument("--hybrid-hot", type=int, default=-1, help="--shards N: the hot-row / tail hybrid (run_hybrid) with this many owner-computed "
                    "popular rows (0: only b and the user node are exact)")
    ap.add_argument("--full-output", action="store_true", help="the full-output block schedule against its B = 1 limit (see full_output_envelope)")
    ap.add_argument("--literal-from", default="", help="--full-output: take the literal (batch_users = 1) curves from this earlier envelope file")
    ap.add_argument("--literal-epochs", type=int, default=0, help="--full-output without a fixture: epochs of the HIP batch_users = 1 run")
    args = ap.parse_args()
    lt = cdae_amd.CROSS_ENTROPY if args.loss == "CE" else cdae_amd.SQUARE
    if args.full_output:
        return full_output_envelope(args, lt)
    fx = fixtures(args.shape, args.num_dim, args.loss, args.seeds, args.fixture_tag)
    if not fx:
        raise SystemExit("no fixture found (tests/golden/make_literal_curves.py makes them)")
    for f in fx:
        seed = int(f["seed"])
        dseed = int(f["data_seed"]) if "data_seed" in f.files else seed
        d = synth.generate_shape(args.shape, seed=dseed)
        ne = int(f["eval_users"]) if "eval_users" in f.files else d.num_users
        ep = args.epochs or len(f["recall10"])
        ref_r, ref_l = f["recall10"][:ep], f["train_loss"][:ep]
        print(json.dumps({"run": "fixture literal", "seed": seed, "recall10": [round(float(x), 5) for x in ref_r],
                          "loss": [round(float(x), 1) for x in ref_l]}), flush=True)
        for shards in args.shards:
            for period in (args.period if shards > 1 else [0]):
                for B in args.batch_users:
                    if shards > 1 and args.relay_epochs:
                        for relay in args.relay_epochs:
                            for combine in args.combine:
                                rec, loss, ups = run_schedule(d, seed, args.num_dim, lt, B, ep, shards, period, relay, combine, ne)
                                dr = np.array(rec) - ref_r
                                dl = np.array(loss) / ref_l - 1.0
                                print(json.dumps({"run": "hip schedule", "shape": args.shape, "seed": seed, "shards": shards, "period": period, "combine": combine,
                                                  "relay_epochs": relay, "batch_users": B, "recall10": [round(x, 5) for x in rec],
                                                  "d_recall": [round(float(x), 5) for x in dr], "max_abs_d_recall_after_relay": round(float(np.abs(dr[int(np.ceil(relay)):]).max()), 5) if int(np.ceil(relay)) < len(dr) else None,
                                                  "rel_d_loss": [round(float(x), 4) for x in dl], "users_per_s": round(ups)}), flush=True)
                        continue
                    if shards > 1 and args.hybrid_hot >= 0:
                        rec, loss, ups = run_hybrid(d, seed, args.num_dim, lt, B, ep, shards, args.hybrid_hot)
                    elif shards == 1:
                        rec, loss, ups = run_single(d, seed, args.num_dim, lt, B, ep, ne=ne)
                    elif args.rule == 0:
                        rec, loss, ups = run_multi(d, seed, args.num_dim, lt, B, ep, shards, period, args.warm_epochs)
                    else:
                        rec, loss, ups = run_sharded(d, seed, args.num_dim, lt, B, ep, shards, period, args.rule)
                    dr = np.abs(np.array(rec) - ref_r)
                    dl = np.array(loss) / ref_l - 1.0
                    print(json.dumps({"run": "hip", "seed": seed, "shards": shards, "period": period, "rule": args.rule, "hybrid_hot": args.hybrid_hot, "warm_epochs": args.warm_epochs if shards > 1 else 0, "batch_users": B,
                                      "recall10": [round(x, 5) for x in rec], "abs_d_recall": [round(float(x), 5) for x in dr],
                                      "max_abs_d_recall": round(float(dr.max()), 5), "rel_d_loss": [round(float(x), 4) for x in dl],
                                      "users_per_s": round(ups)}), flush=True)


if __name__ == "__main__":
    main()
