#!/usr/bin/env python
"""bench.py — users/sec of the CDAE training hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] — ML-10M-shape synthetic (70K users x 10.6K items,
~10M interactions, 80/20 per-user split), K=200, num_neg=5, sigmoid hidden, cross-entropy loss, AdaGrad.
A step = one pass of the hot path (sample -> sort -> encode -> row-major decode -> hidden -> input rows)
over one batch of `batch_users` users, cycling through the shard; with N > 1 every rank (one process per
GPU, launched by torch.distributed.run) trains its OWN ML-10M-shaped data set (weak scaling: per-GPU work
is fixed; --scaling strong shards ONE data set instead) and the ranks exchange the accumulated deltas of the
shared parameters with one RCCL all-reduce per period on a communicator the library owns (cdae_multi.hip).
Inputs (CSR, parameters) are resident in HBM before the timed region.  Timing: barrier +
torch.cuda.synchronize() on both sides of exactly K steps, max over ranks; rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import os as _os
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The library runs three streams (main, prep, aux);
# with RCCL's own streams on top two of them end up sharing a queue and the prep / main overlap is lost (measured: 158 ->
# 254 us per step as soon as an RCCL communicator exists before the handle is created).  Must be set before HIP initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# users per parameter snapshot of the default line.  Chosen from the accuracy envelope, not for speed:
# tests/test_gpu_accuracy.py trains at THIS value against the literal-schedule fixtures of six seeds and asserts a mean Recall@10
# difference within +-0.002 (per seed within the literal schedule's own stream-seed spread, 0.005); DESIGN.md §2 has the sweep
DEFAULT_BATCH_USERS = 256


def algorithmic_bytes_per_user(K, n_u, n_in, num_neg):
    """A_u = 4K[n_in + 4(n_u+m_u) + 4] + 4(n_in+n_u+m_u) + 16(n_u+m_u)."""
    m_u = n_u * num_neg
    return 4.0 * K * (n_in + 4.0 * (n_u + m_u) + 4.0) + 4.0 * (n_in + n_u + m_u) + 16.0 * (n_u + m_u)


def decode_bytes_per_example(K):
    """The decode kernel's share of A_u: per (user, output item) the reference streams the decoder row for
    the dot and re-streams W, W_ag for the AdaGrad step (4 row streams of 4K bytes), plus the item id
    (4 B) and b', b'_ag read+write (16 B)."""
    return 4.0 * K * 4.0 + 4.0 + 16.0


def compulsory_decode_bytes(K, Kp, num_items, examples, batch_users):
    """HBM bytes ONE decode launch must move under the transposed schedule: every item row that has an example
    is read once (D, D_ag), written once (D, D_ag) and copied once to the batch-start snapshot D0 — 5 row streams of 4 Kp bytes,
    b' / b'_ag r+w — the batch's z rows are read once from HBM (every later read is an L2 hit), the sorted example words are
    read (8 B) and one g is written (4 B) per example.  The reference's formulation (4 row streams per
    EXAMPLE, 3220 B each at K=200) is what this schedule avoids; it is reported separately as `reference_algorithmic_bytes`."""
    rows = num_items * (1.0 - np.exp(-examples / num_items))      # rows with at least one of the batch's examples
    return rows * (5.0 * 4.0 * Kp + 16.0) + batch_users * 4.0 * Kp + examples * 12.0


def compulsory_gather_bytes(Kp, num_items, examples, units):
    """HBM bytes the hidden-gradient gather adds when it runs inside the decode launch (decode_gather_kernel, round 6): the D0 rows come
    back from memory once (each XCD's L2 then holds its eighth of them: later reads are L2 hits), 12 B of (item, g, correction id)
    per example, and one partial row per (work unit, item partition) is written."""
    rows = num_items * (1.0 - np.exp(-examples / num_items))
    return rows * 4.0 * Kp + examples * 12.0 + 8.0 * units * 4.0 * Kp


class Watchdog:
    """N > 1 only.  `with WATCHDOG.stage("what", seconds):` around every step that can wait on ANOTHER rank (communicator set-up,
    collectives, barriers): if it does not return in time this rank prints one parseable JSON line with "error" and exits with
    status 4 instead of hanging until the driver's own limit kills the job (a rank that never reaches ncclCommInitRank, a
    collective whose peer died).  One timer thread per armed stage; nothing on the hot path (steps between two syncs are queued
    asynchronously and never armed)."""

    def __init__(self):
        self.rank, self.world, self.args = 0, 1, None

    def configure(self, rank, world, args):
        self.rank, self.world, self.args = rank, world, args

    def stage(self, what, seconds):
        import contextlib
        import threading
        if self.world <= 1:
            return contextlib.nullcontext()
        wd = self

        class _Stage:
            def __enter__(self_):
                self_.t = threading.Timer(seconds, wd.fire, args=(what, seconds))
                self_.t.daemon = True
                self_.t.start()

            def __exit__(self_, *exc):
                self_.t.cancel()
                return False
        return _Stage()

    def fire(self, what, seconds):
        a = self.args
        line = {"metric": "users/sec (whole node)", "value": None, "unit": "users/s", "n_gpus": self.world, "steps": getattr(a, "steps", None),
                "warmup": getattr(a, "warmup", None), "error": f"rank {self.rank}: '{what}' did not complete within {seconds} s (watchdog): "
                "a peer rank is missing or a collective cannot complete", "higher_is_better": True}
        sys.stdout.write(json.dumps(line) + "\n")
        sys.stdout.flush()
        os._exit(4)


WATCHDOG = Watchdog()
# limits (seconds): rendezvous + communicator set-up, and one synchronisation of the whole job (a sync drains up to a few hundred queued
# steps of ~0.1-10 ms each, plus the first collective's lazy RCCL connect)
WD_INIT_S = float(os.environ.get("CDAE_BENCH_WATCHDOG_INIT_S", 30))
WD_SYNC_S = float(os.environ.get("CDAE_BENCH_WATCHDOG_SYNC_S", 120))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=548)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--batch-users", type=int, default=int(os.environ.get("CDAE_BATCH_USERS", DEFAULT_BATCH_USERS)),
                    help="users per parameter snapshot; the default is the largest value with no systematic Recall@10 offset against the "
                         "sequential reference (mean over six fixture seeds within +-0.002, tests/test_gpu_accuracy.py)")
    ap.add_argument("--shape", default="ml10m")
    ap.add_argument("--num-dim", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-users", type=int, default=40000, help="users in the timed CPU-baseline sample (~20 s)")
    ap.add_argument("--seed", type=int, default=20141119)
    ap.add_argument("--profile-every", type=int, default=8, help="HIP-event kernel timing on every n-th batch (0 = off)")
    ap.add_argument("--full-output", action="store_true", help="BASELINE configs[1]/[4]: every unrated item is a negative; "
                    "dense decode on the bf16 MFMA cores (roofline bound: mfma)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1.  weak (default): every rank trains its OWN data set of the named shape (per-GPU work fixed).  strong: "
                         "ONE data set of the named shape, users sharded over the ranks by interactions (BASELINE configs[3]: "
                         "--shape netflix --scaling strong).  Either way the ranks exchange shared-parameter deltas, which is "
                         "OUTSIDE the single-GPU accuracy envelope (DESIGN.md §7): the N > 1 value is a throughput figure")
    ap.add_argument("--layout", choices=["certified", "users", "item-rows"], default=None,
                    help="How N > 1 GPUs divide the model.  certified (the DEFAULT for N > 1 since round 6): the users are sharded, ONE process "
                         "(rank 0) drives all N GPUs through cdae_hip_multi_* on the schedule whose accuracy bounds driver-run tests assert — "
                         "one relayed epoch on the single-GPU schedule (untimed warm-up here; 1 of an application's 50 epochs), then "
                         "synchronous exchanged steps of 64 users per GPU folded in by the global-accumulator rule (DESIGN.md §7); the line "
                         "carries the single-GPU and the item-rows figure of the same node beside it in `config` (--no-side-figures skips "
                         "them).  item-rows: the GPUs cut the ITEM rows of W / b' and the "
                         "decode over them, every GPU sees every user, the user node is sharded by user; two [batch x K] all-reduces per "
                         "batch; it is the single-GPU schedule EXACTLY, so the N > 1 number carries the single-GPU accuracy claim "
                         "(tests/test_gpu_accuracy.py::test_item_rows_sampled_layout_holds_the_accuracy_bounds_at_ml10m_shape).  One "
                         "process drives all N GPUs through cdae_hip_multi_* (rank 0 when launched by torch.distributed.run; the other "
                         "ranks only keep the barriers).  With --full-output it is BASELINE configs[4]'s layout (--shape cfg5_items "
                         "--num-dim 512).  users: user shards + exchange of shared-parameter deltas (one process per GPU, library-owned "
                         "RCCL): scales in throughput but is OUTSIDE the accuracy envelope (DESIGN.md §7) — a throughput figure only")
    ap.add_argument("--no-side-figures", action="store_true", help="--layout certified: do not measure the single-GPU and item-rows figures beside the line")
    ap.add_argument("--sync-batch-users", type=int, default=64, help="--layout certified: users per GPU and exchanged step (64 is the certified size)")
    ap.add_argument("--item-rows-devices", type=int, default=0, help="(internal: the side-figure run of --layout certified) item-rows over this many GPUs from one process")
    ap.add_argument("--users", type=int, default=0, help="--layout item-rows: generate this many users instead of the shape's own count "
                    "(same items and interactions per user), e.g. --shape cfg5_items --users 200000 to see the memory per shard")
    ap.add_argument("--logical-shards", type=int, default=0, help="--layout item-rows on ONE GPU: this many logical shards of cuda:0 "
                    "(the all-reduce is a sum kernel) — measures the cost of the phase structure without a second GPU")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (functional test of the N > 1 path on one GPU; "
                    "RCCL refuses two ranks on one device, so the exchange runs as one-rank groups)")
    ap.add_argument("--exchange-every", type=int, default=-1, help="N > 1: batches between exchanges of the shared-parameter "
                    "deltas (pipelined: the all-reduce overlaps the next period).  -1 (default): chosen at start-up so that one "
                    "period of training covers a measured all-reduce; 0: synchronous exchange after every batch")
    ap.add_argument("--combine", choices=["sum", "global-acc"], default="global-acc",
                    help="--layout users: how the all-reduced deltas are folded in (cdae_hip_delta_set_combine).  global-acc (default): one step "
                         "with the AdaGrad accumulator that has seen every rank; sum: the ranks' accumulated steps are added (rounds 1-4)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world

    WATCHDOG.configure(rank, world, args)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the CDAE hot path has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import cdae_amd
    from cdae_amd import synth
    from cdae_amd.distributed import shard_bounds

    if args.layout is None:
        args.layout = "item-rows" if (args.logical_shards or args.item_rows_devices or (world > 1 and args.full_output)) else ("certified" if world > 1 else "users")
    if args.layout == "item-rows":
        return bench_item_rows(args, rank, world)
    if args.layout == "certified" and world > 1 and not (args.logical_shards or args.share_device) and torch.cuda.device_count() < world:
        # rank 0 drives every GPU of the node in this layout: a launcher that shows each rank only its own device cannot run it
        print(f"[bench] --layout certified needs all {world} devices visible to rank 0 (this process sees {torch.cuda.device_count()}): "
              "falling back to --layout users (one process per GPU; THROUGHPUT ONLY, see config.accuracy)", file=sys.stderr)
        args.layout = "users"
    if args.layout == "certified":
        return bench_certified(args, rank, world)
    if args.scaling == "strong" and world > 1:
        whole = synth.generate_shape(args.shape, seed=args.seed)          # the same data set on every rank ...
        u0, u1 = shard_bounds(whole.num_users, world, rank, whole.train_ptr)
        data, uid_offset = whole.user_range(u0, u1), u0                   # ... of which this rank trains its users
        shape_note = f"{args.shape}-shape synthetic {whole.num_users}x{whole.num_items} sharded over {world} GPUs by interactions"
        del whole
    else:
        # every rank owns one data set of the named shape over the same item space
        data, uid_offset = synth.generate_shape(args.shape, seed=args.seed + 7919 * rank), rank * synth.SHAPES[args.shape][0]
        shape_note = f"{args.shape}-shape synthetic {data.num_users}x{data.num_items} per GPU"
    K, B = args.num_dim, min(args.batch_users, data.num_users)
    cfg = cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, num_neg=5, num_corruptions=1,
                              corruption_ratio=0.5, scaled=True, learn_rate=0.1, beta=1.0, lambda_=0.01,
                              using_adagrad=True, user_factor=True, batch_users=B, full_output=args.full_output)
    model = cdae_amd.CDAE(cfg, device=local_rank)
    model.set_interactions(data.num_users, data.num_items, data.train_ptr, data.train_col, user_id_offset=uid_offset)
    model.init_params(args.seed)         # identical shared parameters on every rank; Wu is private (keyed by global user id)
    plan = model.full_output_plan        # CDAE_PLAN_* bits: which launches the full-output decode is made of (0: sampled decode)
    # Rendezvous over gloo (CPU): torch.distributed only carries the 128-byte RCCL id, the barriers and the timing reductions.
    # The data path — one all-reduce of the staged deltas per period — runs inside the library on its own RCCL communicator
    # and stream (cdae_multi.hip), created AFTER the handle's streams: with a communicator in place first the handle's
    # streams share hardware queues with RCCL's and the overlapped exchange is several times slower (r01 measurement).
    dist = None
    force_dist = bool(os.environ.get("CDAE_BENCH_FORCE_DIST"))   # developer aid: exercise the RCCL path with one rank
    exchanging = world > 1 or force_dist
    if exchanging:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        with WATCHDOG.stage("gloo rendezvous + ncclCommInitRank", WD_INIT_S):
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=WD_INIT_S))
            ids = [cdae_amd.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            if os.environ.get("CDAE_BENCH_NO_COMM"):
                pass                                                        # developer aid: the exchange schedule without any RCCL object
            elif args.share_device and world > 1:
                model.comm_init_rank(1, 0, cdae_amd.comm_unique_id())      # functional smoke test: RCCL refuses duplicate devices
            else:
                model.comm_init_rank(world, rank, ids[0])
        model.delta_set_combine(cdae_amd.COMBINE_GLOBAL_ACC if args.combine == "global-acc" else cdae_amd.COMBINE_SUM)
        model.exchange_configure(max(0, args.exchange_every) if args.exchange_every >= 0 else 1 << 30)

    n_batches = (data.num_users + B - 1) // B

    def batch_of(i):
        b = i % n_batches
        return i // n_batches, b * B, min(data.num_users, (b + 1) * B)

    KEYS = ("users", "examples", "batches", "ms_sample", "ms_sort", "ms_encode", "ms_decode", "ms_hidden", "ms_input", "launches_decode")
    acc = {k: 0 for k in KEYS}

    def add(st):
        for k in KEYS:
            acc[k] += getattr(st, k)

    def prefetch_from(i):
        # the library prepares as many leading batches of the range as its look-ahead is deep (one, or two with the second
        # prep lane): hand it the next two batches where they are consecutive in one epoch
        nep, n0, n1 = batch_of(i)
        mep, m0, m1 = batch_of(i + 1)
        if mep == nep and m0 == n1:
            n1 = m1
        model.prefetch_users(args.seed, nep, n0, n1)

    def step(i):
        ep, u0, u1 = batch_of(i)
        # steps queue asynchronously on the library's stream; the next batch is sampled and sorted on the side
        # stream while this one trains (and, with N > 1, while the previous period's deltas are all-reduced)
        model.enqueue_users(args.seed, ep, u0, u1)
        prefetch_from(i + 1)
        if exchanging:
            model.exchange_step()

    def host_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def sync():
        with WATCHDOG.stage("barrier + device synchronisation (queued steps and their collectives)", WD_SYNC_S):
            if dist is not None:
                dist.barrier()
            model.synchronize()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()

    exchange_note = None
    if exchanging and args.exchange_every < 0:
        # auto period: time a few batches without exchange and a few idle all-reduces of the real buffer, before anything
        # that counts has been staged (the period is 2^30 here: no boundary is reached)
        sync()
        n_cal = max(3, min(args.warmup, 10))        # calibration batches (set-up, before the W warm-up steps)
        step(0)                                      # first batch: cold pipeline, not timed
        sync()
        tw = time.perf_counter()
        for i in range(1, 1 + n_cal):
            step(i)
        sync()
        t_step = (time.perf_counter() - tw) / n_cal
        t_ar = model.exchange_time_all_reduce(5)     # flushes what the calibration batches did, then restarts the exchange
        want = int(min(8, max(2, -(-1.5 * t_ar // max(t_step, 1e-9)))))       # a boundary costs ~40 us of stream time: never every batch
        args.exchange_every = int(host_max(want))
        model.exchange_configure(args.exchange_every)
        exchange_note = f"period chosen at start-up: all-reduce {t_ar * 1e6:.0f} us vs {t_step * 1e6:.0f} us per batch"
    def run(first, count):
        # `count` steps from step `first`.  Without an exchange the steps are handed to the library in runs of up to 16
        # consecutive batches of one epoch per call (as cdae_hip_train_epoch would: the batch loop is the library's, in C);
        # with an exchange every step is followed by its exchange_step
        i, end = first, first + count
        while i < end:
            if exchanging:
                step(i)
                i += 1
                continue
            ep, u0, u1 = batch_of(i)
            c = 1
            while c < 16 and i + c < end and batch_of(i + c)[0] == ep and batch_of(i + c)[1] == u1:
                u1 = batch_of(i + c)[2]
                c += 1
            model.enqueue_users(args.seed, ep, u0, u1)
            prefetch_from(i + c)
            i += c

    # Device pre-warm (untimed, BEFORE the W warm-up steps of the contract, disclosed as `prewarm_steps`): the driver's short form
    # (--steps 20 --warmup 5) times 2 ms on a device five batches out of idle, which reads 3-5 % slower than the steady state a
    # training epoch runs in (clocks, caches; DESIGN.md §8: 98 us per step after 5 steps, 94-95 after 300).  The same kind of steps on
    # the same data; the W warm-up steps and the K timed steps follow unchanged.
    prewarm = max(0, 300 - args.warmup)
    # ... and what the pre-warm hides is reported, not dropped: the first COLD_STEPS steps of the process (device out of idle, cold
    # caches, first use of every kernel), timed on their own as `cold_start_ms_per_step`
    COLD_STEPS = min(20, prewarm)
    cold_ms = None
    if COLD_STEPS and not exchanging:
        sync()
        tc = time.perf_counter()
        run(0, COLD_STEPS)
        sync()
        cold_ms = 1e3 * (time.perf_counter() - tc) / COLD_STEPS
        run(COLD_STEPS, prewarm - COLD_STEPS)
    else:
        run(0, prewarm)
    run(prewarm, args.warmup)
    args_first = prewarm + args.warmup
    model.collect_stats()
    acc = {k: 0 for k in KEYS}
    # Timed region: HIP events (on the library's own stream) around the DECODE launch of every `profile_every`-th batch — the
    # roofline's kernel, measured live; an event pair costs ~6 us of stream time, so the other five families (14 records, 42 us
    # on a profiled step) are timed in an untimed pass after it
    period = min(args.profile_every, max(1, args.steps // 5)) if args.profile_every else 0      # short runs: at least ~5 timed launches
    model.set_profiling(period, families=("decode",))
    sync()
    t0 = time.perf_counter()
    run(args_first, args.steps)
    if exchanging:
        model.exchange_flush()            # the last period's deltas are reduced and merged inside the timed region
    t_queued = time.perf_counter()
    sync()
    elapsed = time.perf_counter() - t0
    if os.environ.get("CDAE_BENCH_PHASES"):      # developer aid: where the host spends the timed region
        t_a = time.perf_counter(); model.synchronize(); t_b = time.perf_counter(); torch.cuda.synchronize(); t_c = time.perf_counter()
        print(f"[phases] enqueue {1e6 * (t_queued - t0):.0f} us, wait {1e6 * (elapsed - (t_queued - t0)):.0f} us; on an idle device: "
              f"model.synchronize {1e6 * (t_b - t_a):.1f} us, torch.cuda.synchronize {1e6 * (t_c - t_b):.1f} us", file=sys.stderr)
    add(model.collect_stats())
    timed = dict(acc)
    if args.profile_every:
        # untimed: the other kernel families, every second batch of 32 more steps (kernel_ms_per_step; not part of `value`)
        model.set_profiling(2)
        run(args_first + args.steps, 32)
        if exchanging:
            model.exchange_flush()
        sync()
        extra = model.collect_stats()
        for k in ("ms_sample", "ms_sort", "ms_encode", "ms_hidden", "ms_input"):
            acc[k] = getattr(extra, k) / max(1, extra.launches_decode) * max(1, timed["launches_decode"])   # per profiled step, on the timed pass's scale
    model.set_profiling(False)

    users_total = float(acc["users"])
    if dist is not None:
        t = torch.tensor([users_total], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, users_total = host_max(elapsed), float(t[0])
        # RCCL writes a version banner to the C stdout buffer of every rank: push it out now, on all ranks, so that
        # rank 0's JSON line is the last thing the job prints
        import ctypes
        ctypes.CDLL(None).fflush(None)
        dist.barrier()
    if rank != 0:
        model.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    n_u = data.nnz_train / data.num_users
    n_in = n_u * (1.0 - cfg.corruption_ratio)
    a_user = algorithmic_bytes_per_user(K, n_u, n_in, cfg.num_neg)
    value = users_total / elapsed
    # roofline of the dominant kernel (decode), rank 0's launches
    ex_per_launch = acc["examples"] / max(1, acc["batches"])
    users_per_launch = acc["users"] / max(1, acc["batches"])
    ms_per_launch = acc["ms_decode"] / max(1, acc["launches_decode"])
    Kp = 64 * (1 if K <= 64 else 2 if K <= 128 else 4 if K <= 256 else 8)
    if args.full_output:
        # dominant kernels: the three bf16 MFMA contractions, 6 K I flop per user, timed as one family
        MFMA_PEAK_TFLOPS = 2500.0       # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
        # which of the three products the timed "decode" family holds (cdae_hip_full_output_plan): with the row steps fused into
        # GEMM 3 (K > 256 over >= 32768 items) that launch is HBM-bound, lives in the "input" family, and the decode family holds two
        rows_fused = bool(plan & cdae_amd.binding.PLAN_ROWS_FUSED)
        products = 2 if rows_fused else 3
        flops_launch = 2.0 * products * K * data.num_items * users_per_launch
        achieved_tf = flops_launch / (ms_per_launch * 1e-3) / 1e12 if ms_per_launch > 0 else 0.0
        step_tf = 6.0 * K * data.num_items * users_per_launch / (1e-3 * 1e3 * elapsed / args.steps) / 1e12
        kernels = ("full_decode_fused_kernel + gemm_nt_bf16_lds_kernel (GEMM 3)" if plan & cdae_amd.binding.PLAN_FUSED_DECODE else
                   "gemm_nt_bf16_ldsw_kernel<EPI_LOSS> (GEMM 1) + " + ("gemm_tn_bf16_kernel" if plan & cdae_amd.binding.PLAN_GEMM2_TN else "gemm_nt_bf16_ldsw_kernel")
                   + " (GEMM 2)" + ("; GEMM 3 runs inside gemm3_rows_fused_kernel with the row steps (HBM-bound, 'input' family) and is NOT counted here"
                                    if rows_fused else " + gemm_nt_bf16_ldsw_kernel (GEMM 3)"))
        step_ms = 1e3 * elapsed / args.steps
        if rows_fused:
            # K > 256 over >= 32768 items (BASELINE configs[4]): two timed families, each priced against the LARGER of its two floors —
            # algorithmic flops at the dense bf16 peak, algorithmic HBM bytes at 8 TB/s — and the headline record is the launch that
            # takes most of the step, which is the HBM-bound one (round 3 reported the matrix-core fraction of the other family only).
            I_, Bp_ = float(data.num_items), float(-(-int(users_per_launch) // 256) * 256)
            ms_rows = acc["ms_input"] / max(1, acc["launches_decode"])
            fam = {
                # GEMM 1 (forward + loss', z rows in registers: reads the bf16 image of D, writes G^T) + GEMM 2 (reads G^T and the image)
                "decode": {"kernels": "gemm1_loss_duo_kernel + full_positive_fixup_kernel + gemm_tn_bf16_kernel", "ms": ms_per_launch,
                           "flops": 4.0 * K * I_ * users_per_launch, "bytes": 2.0 * (2.0 * I_ * Kp) + 2.0 * (2.0 * I_ * Bp_)},
                # GEMM 3 + the row steps: D and D_ag read and written once (fp32), the bf16 image written, G^T read once
                "rows": {"kernels": "gemm3_rows_fused_kernel + full_rows_inputs_kernel", "ms": ms_rows,
                         "flops": 2.0 * K * I_ * users_per_launch, "bytes": I_ * Kp * (4 * 4 + 2) + 2.0 * I_ * Bp_},
            }
            for f in fam.values():
                f["mfma_floor_ms"] = f["flops"] / (MFMA_PEAK_TFLOPS * 1e12) * 1e3
                f["hbm_floor_ms"] = f["bytes"] / (HBM_PEAK_GBS * 1e9) * 1e3
                f["bound"] = "mfma" if f["mfma_floor_ms"] >= f["hbm_floor_ms"] else "hbm"
                f["frac"] = max(f["mfma_floor_ms"], f["hbm_floor_ms"]) / f["ms"] if f["ms"] > 0 else 0.0
                f["achieved_tflops"] = f["flops"] / (f["ms"] * 1e-3) / 1e12 if f["ms"] > 0 else 0.0
                f["achieved_gbs"] = f["bytes"] / (f["ms"] * 1e-3) / 1e9 if f["ms"] > 0 else 0.0
            fam["decode"]["launches"], fam["rows"]["launches"] = 2, 1       # (the decode family is two launches of about equal length: GEMM 1, GEMM 2)
            dom = max(fam.values(), key=lambda f: f["ms"] / f["launches"])   # the dominant LAUNCH, not the longest family
            traffic, traffic_source = measured_full_traffic(args.shape, K, B)
            floors = sum(max(f["mfma_floor_ms"], f["hbm_floor_ms"]) for f in fam.values())
            roofline = {"bound": dom["bound"], "kernel": dom["kernels"],
                        "achieved": dom["achieved_gbs"] if dom["bound"] == "hbm" else dom["achieved_tflops"],
                        "peak": HBM_PEAK_GBS if dom["bound"] == "hbm" else MFMA_PEAK_TFLOPS,
                        "unit": "GB/s" if dom["bound"] == "hbm" else "TFLOP/s", "frac": dom["frac"],
                        "traffic": traffic, "traffic_source": traffic_source,
                        "frac_definition": "the dominant launch's binding floor (max of algorithmic flops at the dense bf16 peak and algorithmic HBM bytes at 8 TB/s) / its measured time",
                        "avg_launch_ms": dom["ms"], "per_launch": fam,
                        "whole_step": {"achieved": step_tf, "frac": step_tf / MFMA_PEAK_TFLOPS, "binding_floor_frac": floors / step_ms,
                                       "note": "frac: all three products' 6 K I flop per user over the whole step at the bf16 peak; binding_floor_frac: the sum "
                                               "of the families' binding floors / the whole step (encode, hidden layer and every launch boundary included)"}}
        else:
            hbm_bytes = 2.0 * data.num_items * Kp * 2 + 2.0 * data.num_items * max(128.0, users_per_launch) * 2      # bf16 images of D / D^T in, G^T out and in
            roofline = {"bound": "mfma", "kernel": kernels + " (+ positive fix-up, rated-items bitmap)",
                        "achieved": achieved_tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tf / MFMA_PEAK_TFLOPS,
                        "traffic": None, "algorithmic_flops_per_launch": flops_launch, "products_in_family": products,
                        "avg_launch_ms": ms_per_launch,
                        "hbm_floor_ms": hbm_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, "mfma_floor_ms": flops_launch / (MFMA_PEAK_TFLOPS * 1e12) * 1e3,
                        "note": "small item spaces: the step is a chain of 5-20 us launches — neither floor binds (DESIGN.md §5b)",
                        "whole_step": {"achieved": step_tf, "frac": step_tf / MFMA_PEAK_TFLOPS,
                                       "note": "all three products' 6 K I flop per user over the whole step (encode, row steps and every launch boundary included)"}}
        workload = f"{shape_note}, nnz_train={data.nnz_train}, K={K}, FULL-OUTPUT decode (every unrated item a negative), CE loss, AdaGrad, q=0.5 scaled"
    else:
        # HBM roofline on the bytes the launch MUST move (never above 1); what actually bounds the kernel is stated beside it
        comp = compulsory_decode_bytes(K, Kp, data.num_items, ex_per_launch, users_per_launch)
        dplan = model.decode_plan if hasattr(model, "decode_plan") else dict(hot_rows=0, late_rows=0, fused=False)
        if dplan["fused"]:
            # the launch also gathers the hidden gradient (one launch since round 6): its compulsory bytes ride along
            unit_pos = 64 if B <= 1024 else 128
            units_per_user = float(np.ceil(np.diff(data.train_ptr) / unit_pos).sum()) / data.num_users
            comp += compulsory_gather_bytes(Kp, data.num_items, ex_per_launch, units_per_user * users_per_launch)
        achieved = comp / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
        traffic, traffic_source = measured_traffic(args.shape, K, B)
        top_share = float(np.bincount(data.train_col, minlength=data.num_items).max()) / data.num_users
        chain = users_per_launch * top_share * (1.0 + 0.05)            # positives of the most popular row + its few negatives
        # the two instruction-issue bounds, as fitted to measurement (CDAE_DEBUG_SKIP_ROLES 8 / 4 = the popular rows /
        # the four-rows-per-wavefront body alone, at 256 and 512 users per batch): a lone wavefront walks the most popular row at
        # ~420 cycles per example after ~5 us of launch + prologue (28.6 / 50.3 us alone); the four-row body costs ~1350 SIMD cycles
        # per step of four examples (121 VALU + 12 scalar instructions, DPP wait states, ~2.6 wavefronts sharing a SIMD: 32-34 us
        # alone at 256 users), spread over 1024 SIMDs with the groups of a wavefront ~85 % full
        CYC_PER_EXAMPLE, CLOCK_GHZ = 420.0, 2.4
        chain_us = 5.0 + chain * CYC_PER_EXAMPLE / (CLOCK_GHZ * 1e3)
        issue_us = 4.0 + (ex_per_launch / 4.0 / 0.85) * 1350.0 / (1024 * CLOCK_GHZ * 1e3)
        roofline = {"bound": "hbm", "kernel": "decode_gather_kernel (decode + hidden-gradient gather, one launch)" if dplan["fused"] else "decode_hybrid_kernel",
                    "decode_plan": dplan, "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    # `traffic` is NOT measured by this run (a process cannot collect PMC counters on itself): it is read back from
                    # the committed rocprofv3 --pmc passes of this same command, when one exists for this exact workload
                    "traffic_source": traffic_source,
                    "frac_definition": "compulsory bytes of the transposed schedule / launch time / HBM peak (DESIGN.md §5)",
                    # SURVEY.md §8(d)'s own figure — the REFERENCE formulation's four row streams per example (3220 B at K=200) — over
                    # the same launch time: above 1, because the schedule keeps each row in registers for the batch and never moves
                    # those bytes.  Reported so both definitions are explicit; it is a speed-up proxy, not a roofline fraction.
                    "survey_8d_frac": (decode_bytes_per_example(K) * ex_per_launch / (ms_per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS
                                       if ms_per_launch > 0 else None),
                    "compulsory_bytes_per_launch": comp, "avg_launch_ms": ms_per_launch,
                    "note": "the kernel is NOT bandwidth-bound: rows stay in registers for the whole batch; see `other_bounds`",
                    "other_bounds": {"top_row_serial_chain_us": chain_us,
                                     "four_rows_per_wave_issue_us": issue_us,
                                     "frac_of_launch_explained_by_larger": max(chain_us, issue_us) / (ms_per_launch * 1e3) if ms_per_launch > 0 else None},
                    "reference_algorithmic_bytes_per_launch": decode_bytes_per_example(K) * ex_per_launch,
                    "whole_step_users_per_s_over_reference_hbm_roof": value / args.gpus * a_user / 1e9 / HBM_PEAK_GBS}
        workload = f"{shape_note}, nnz_train={data.nnz_train}, K={K}, num_neg=5, CE loss, AdaGrad, q=0.5 scaled"
    if not exchanging:
        exchange = "none"
    elif args.exchange_every > 0:
        exchange = (f"library-owned RCCL all-reduce(sum) of shared-parameter deltas every {args.exchange_every} batches, pipelined "
                    f"(merged one period late)" + (f" ({exchange_note})" if exchange_note else ""))
    else:
        exchange = "library-owned RCCL all-reduce(sum) of shared-parameter deltas after every batch (synchronous)"
    out = {
        # BASELINE.json's metric on its own workload; other shapes / K (developer runs) are named as what they are
        "metric": ("users/sec (whole node) K=200 ML-10M-shape; Recall@10 parity" if (args.shape == "ml10m" and args.num_dim == 200)
                   else f"users/sec (whole node) K={args.num_dim} {args.shape}-shape"),
        "value": value, "unit": "users/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling if args.gpus > 1 else "weak",
        "vs_baseline": None, "dtype": "bf16" if args.full_output else "f32", "data": "synthetic",
        "config": {"workload": workload, "batch_users": B, "global_batch": B * args.gpus, "parallelism": f"dp{args.gpus}",
                   "exchange": exchange,
                   "accuracy": (full_output_accuracy(B, data.num_users, args.shape, K) if args.full_output and args.gpus == 1
                                else "batch_users within the single-GPU envelope of tests/test_gpu_accuracy.py" if args.gpus == 1 and B <= DEFAULT_BATCH_USERS
                                else "single GPU, batch_users ABOVE the accuracy envelope (throughput only)" if args.gpus == 1
                                else ("THROUGHPUT ONLY: this multi-process form (--layout users: one process per GPU, synchronous %s exchange) runs NO relayed "
                                      "epoch — the relay exists only in cdae_hip_multi_set_schedule, i.e. in --layout certified, whose line carries the accuracy "
                                      "bounds (DESIGN.md §7)" % args.combine) if args.exchange_every == 0 and B <= 64
                                else "data-parallel delta exchange (pipelined, or more than 64 users per rank and step): OUTSIDE the measured envelope "
                                     "(DESIGN.md §7 table); throughput only")},
        "roofline": roofline,
        "kernel_ms_per_step": {k[3:]: acc[k] / max(1, acc["launches_decode"]) for k in acc if k.startswith("ms_")},
        "profiled_steps": int(acc["launches_decode"]),
        "prewarm_steps": prewarm,          # untimed device pre-warm in front of the W warm-up steps (see the comment at `prewarm`)
        "cold_start_ms_per_step": cold_ms, # the process's first 20 steps (part of the pre-warm), timed on their own
        "kernel_ms_note": "decode: HIP events inside the timed region; the other families: an untimed pass of 32 steps after it",
    }
    if not args.no_cpu_baseline and args.gpus == 1:          # reported at N = 1 only (rank 0's host cores)
        out["cpu_baseline"] = cpu_baseline(data, cfg, args)
    model.close()
    if dist is not None:
        dist.destroy_process_group()
        import ctypes
        ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


def bench_item_rows(args, rank, world):
    """--layout item-rows: ONE process (rank 0) drives all N GPUs through cdae_hip_multi_* with CDAE_LAYOUT_ITEM_ROWS — the sampled
    decode (BASELINE configs[2] / [3] on N GPUs, the default for N > 1) or, with --full-output, the bf16 matrix-core decode
    (configs[4]).  One data set, strong scaling; a step = one batch of `batch_users` users through all three phases on every GPU."""
    import torch
    import cdae_amd
    from cdae_amd import synth
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with WATCHDOG.stage("gloo rendezvous", WD_INIT_S):
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank != 0:                        # the other ranks own no GPU work in this layout: they keep the job's barriers
        dist.barrier(); dist.barrier()
        dist.destroy_process_group()
        return
    devices = [0] * args.logical_shards if args.logical_shards else list(range(args.item_rows_devices or world))
    if args.users >= 2_000_000:          # scale run (configs[4]'s 10 M users): 20 interactions per user, uniform over the items
        u, i, nnz = synth.SHAPES[args.shape]
        t_gen = time.perf_counter()
        data = synth.generate_uniform(args.users, i, per_user=20, seed=args.seed)
        print(f"[bench] {args.users} users x 20 stratified-uniform interactions generated in {time.perf_counter() - t_gen:.1f} s", file=sys.stderr)
    elif args.users:
        u, i, nnz = synth.SHAPES[args.shape]
        data = synth.generate(args.users, i, int(nnz * (args.users / u)), seed=args.seed)
    else:
        data = synth.generate_shape(args.shape, seed=args.seed)
    K, B = args.num_dim, min(args.batch_users, data.num_users)
    free0 = torch.cuda.mem_get_info(0)[0]
    cfg = cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, num_neg=5, num_corruptions=1, corruption_ratio=0.5, scaled=True,
                              learn_rate=0.1, beta=1.0, lambda_=0.01, using_adagrad=True, user_factor=True, batch_users=B,
                              full_output=args.full_output)
    model = cdae_amd.MultiCDAE(cfg, devices=devices, item_rows=True)
    model.reset(data, seed=args.seed)
    torch.cuda.synchronize()
    used_gib = (free0 - torch.cuda.mem_get_info(0)[0]) / 2**30         # device 0: all logical shards, or shard 0 of a multi-GPU run
    n_batches = (data.num_users + B - 1) // B
    RUN = 1 if args.full_output else 16  # sampled decode: a step is ~0.1 ms, so batches are handed over in runs (one host sync per run)

    prof = {"ms_decode": 0.0, "ms_input": 0.0, "launches": 0}      # shard 0's HIP-event kernel times (filled while its profiling is on)

    def run(first, count):
        """batches first .. first + count - 1 (cycling through the data set, epoch = pass number); returns users trained"""
        users, i, end = 0, first, first + count
        while i < end:
            b = i % n_batches
            n = min(RUN, end - i, n_batches - b)
            st = model.train_users(args.seed, i // n_batches, b * B, min(data.num_users, (b + n) * B))
            users += st.users
            prof["ms_decode"] += st.ms_decode; prof["ms_input"] += st.ms_input; prof["launches"] += int(st.launches_decode)
            i += n
        return users

    run(0, args.warmup)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    users = run(args.warmup, args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    # per-kernel record (untimed pass behind the timed region): HIP events around shard 0's decode / row launches of a few batches
    prof.update(ms_decode=0.0, ms_input=0.0, launches=0)
    model.shard_profiling(0, 1, [3, 5])
    run(args.warmup + args.steps, min(8, args.steps))
    model.shard_profiling(0, 0)
    k_launches = max(1, prof["launches"])
    ms_decode0, ms_rows0 = prof["ms_decode"] / k_launches, prof["ms_input"] / k_launches
    i0, i1 = model.shards()[0]
    n_gpus = len(set(devices))
    Kp = 64 * (1 if K <= 64 else 2 if K <= 128 else 4 if K <= 256 else 8)
    par = f"item-rows x{len(devices)}" + (" (logical shards of one GPU)" if args.logical_shards else "")
    if args.full_output:
        MFMA_PEAK_TFLOPS = 2500.0
        flops_step = 6.0 * K * data.num_items * B
        achieved = flops_step * args.steps / elapsed / 1e12 / n_gpus          # per GPU, whole step (not the decode family alone)
        metric = f"users/sec (whole node) K={K} {args.shape}-shape full-output, item-rows layout"
        work = "FULL-OUTPUT decode, CE loss, AdaGrad, q=0.5 scaled"
        accuracy = ("the single-GPU full-output schedule exactly (tests/test_gpu_multi.py: parameters within 5e-3 of range of the single handle); that schedule: "
                    + full_output_accuracy(B, data.num_users, args.shape, K))
        I0, Bp0 = float(i1 - i0), float(-(-B // 256) * 256)
        shard0 = {"item_rows": int(i1 - i0), "decode_family_ms": ms_decode0, "row_family_ms": ms_rows0,
                  "decode_mfma_floor_ms": 4.0 * K * I0 * B / (MFMA_PEAK_TFLOPS * 1e12) * 1e3,
                  "decode_hbm_floor_ms": (4.0 * I0 * Kp + 4.0 * I0 * Bp0) / (HBM_PEAK_GBS * 1e9) * 1e3,
                  "rows_mfma_floor_ms": 2.0 * K * I0 * B / (MFMA_PEAK_TFLOPS * 1e12) * 1e3,
                  "rows_hbm_floor_ms": (I0 * Kp * 18.0 + 2.0 * I0 * Bp0) / (HBM_PEAK_GBS * 1e9) * 1e3,
                  "note": "shard 0's own launches over its item rows (HIP events, untimed pass): forward + loss' + hidden-gradient products ('decode') and "
                          "GEMM 3 + row steps ('row'), each beside its two floors (algorithmic flops at the bf16 peak, algorithmic bytes at 8 TB/s)"}
        roofline = {"bound": "mfma", "kernel": "whole step per GPU (three bf16 products over the local item rows + row steps + replicated hidden layer)",
                    "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": None,
                    "shard0_launches": shard0}
    else:
        ex_per_batch = data.nnz_train * 6.0 / n_batches
        comp = compulsory_decode_bytes(K, Kp, data.num_items, ex_per_batch, B) / n_gpus      # every GPU moves its share of the rows once
        achieved = comp / (elapsed / args.steps) / 1e9
        metric = ("users/sec (whole node) K=200 ML-10M-shape; Recall@10 parity" if (args.shape == "ml10m" and K == 200)
                  else f"users/sec (whole node) K={K} {args.shape}-shape")
        work = "num_neg=5, CE loss, AdaGrad, q=0.5 scaled"
        accuracy = ("the single-GPU schedule exactly: per-row chains sequential over the GLOBAL batch, two all-reduced per-user sums "
                    "(tests/test_gpu_multi.py: one shard == the single handle bit for bit; tests/test_gpu_accuracy.py: four shards hold "
                    "the single-GPU Recall@10 / loss bounds at ML-10M shape)" if B <= DEFAULT_BATCH_USERS
                    else "the single-GPU schedule exactly, at a batch_users ABOVE the accuracy envelope (throughput only)")
        roofline = {"bound": "hbm", "kernel": "whole step per GPU (the decode launch is not timed on its own in this layout)",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "frac_definition": "this GPU's share of the decode's compulsory bytes / WHOLE step time / HBM peak — a lower bound of the "
                                       "decode kernel's own fraction; the step is bound by two latency-bound all-reduces and the longest row "
                                       "chain, not by bandwidth (DESIGN.md §7b cost model)",
                    # the decode launch of shard 0 on its own (HIP events, untimed pass behind the timed region): its rows' compulsory bytes / its time
                    "shard0_decode": {"item_rows": int(i1 - i0), "avg_launch_ms": ms_decode0,
                                      "compulsory_bytes": compulsory_decode_bytes(K, Kp, int(i1 - i0), ex_per_batch * (i1 - i0) / data.num_items, B),
                                      "frac": (compulsory_decode_bytes(K, Kp, int(i1 - i0), ex_per_batch * (i1 - i0) / data.num_items, B) / (ms_decode0 * 1e-3) / 1e9 / HBM_PEAK_GBS
                                               if ms_decode0 > 0 else None),
                                      "note": "examples of shard 0 estimated as its share of the item rows (negatives are uniform over items)"}}
    out = {"metric": metric,
           "value": users / elapsed, "unit": "users/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "bf16" if args.full_output else "f32", "data": "synthetic",
           "config": {"workload": f"{args.shape}-shape synthetic {data.num_users}x{data.num_items} (ONE data set, item rows cut over the GPUs), "
                                  f"nnz_train={data.nnz_train}, K={K}, {work}",
                      "batch_users": B, "global_batch": B, "parallelism": par,
                      "exchange": "two all-reduces per batch: [batch_users x row_stride] fp32 input sums (+ the owners' Wu rows) and hidden "
                                  "gradient; no item-row parameter crosses GPUs, the user node is sharded by user",
                      "accuracy": accuracy},
           "roofline": roofline,
           # what the library allocated on device 0 (all logical shards when --logical-shards, else shard 0): item rows + their
           # workspaces, this shard's share of the user node (Wu / Wu_ag are sharded by user range), example buffers
           "device0_memory_gib": round(used_gib, 3), "shards_on_device0": len(devices) if args.logical_shards else 1,
           "user_node_gib_per_shard": round(2.0 * (data.num_users / len(devices)) * Kp * 4 / 2**30, 3),
           "user_node_gib_if_replicated": round(2.0 * data.num_users * Kp * 4 / 2**30, 3)}
    model.close()
    if dist is not None:
        dist.destroy_process_group()
    print(json.dumps(out), flush=True)


def side_figure(extra, timeout_s=240):
    """`value` of a short run of this script with other flags, in a fresh process that sees every GPU of the node and no
    torch.distributed environment (the side figures of --layout certified); None (and the reason) when it fails"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                                                             "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-side-figures"] + extra, env=env,
                             capture_output=True, text=True, timeout=timeout_s)
        line = json.loads(out.stdout.strip().splitlines()[-1])
        return {"users_per_s": line["value"], "ms_per_step": line["ms_per_step"], "batch_users": line["config"].get("batch_users"), "command": "bench.py " + " ".join(extra)}
    except Exception as e:                                  # noqa: BLE001  (a side figure must never take the line down)
        return {"users_per_s": None, "error": repr(e)[:200], "command": "bench.py " + " ".join(extra)}


def bench_certified(args, rank, world):
    """--layout certified (the default for N > 1): the user-sharded layout on the schedule whose accuracy bounds are asserted by driver-run
    tests (tests/test_gpu_accuracy.py::test_relay_then_exchange_schedule_on_eight_shards_at_ml10m_shape, tests/test_gpu_netflix.py::
    test_netflix_relay_then_exchange_schedule_on_eight_shards): relay epoch, then synchronous exchanged steps of --sync-batch-users users per
    GPU under the global-accumulator combine rule.  ONE process (rank 0) drives all N GPUs through cdae_hip_multi_* (a host thread per GPU
    inside the library, RCCL communicators of ncclCommInitAll); a step = one exchanged step = N x sync_batch_users users.  ONE data set
    (strong scaling).  The relayed epoch runs before the timed region: it costs one single-GPU epoch once per training run."""
    import torch
    import cdae_amd
    from cdae_amd import synth
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with WATCHDOG.stage("gloo rendezvous", WD_INIT_S):
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank != 0:                        # the other ranks own no GPU work: they keep the job's barriers
        dist.barrier(); dist.barrier()
        dist.destroy_process_group()
        return
    n_shards = args.logical_shards or world
    devices = [0] * n_shards if (args.logical_shards or args.share_device) else list(range(world))
    side = {}
    if not args.no_side_figures:
        # the same node's other two answers to "N GPUs", measured now, in their own processes, before this one holds the devices
        base = ["--shape", args.shape, "--num-dim", str(args.num_dim), "--seed", str(args.seed)]
        side["single_gpu"] = side_figure(base + ["--steps", "200", "--warmup", "40"])
        if len(set(devices)) > 1:
            side["item_rows_same_gpus"] = side_figure(base + ["--layout", "item-rows", "--item-rows-devices", str(world), "--steps", "100", "--warmup", "20"])
        else:
            side["item_rows_same_gpus"] = side_figure(base + ["--layout", "item-rows", "--logical-shards", str(n_shards), "--steps", "60", "--warmup", "10"])
    data = synth.generate_shape(args.shape, seed=args.seed)
    K, B1 = args.num_dim, min(args.batch_users, data.num_users)
    sb = max(1, min(args.sync_batch_users, B1))
    cfg = cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, num_neg=5, num_corruptions=1, corruption_ratio=0.5, scaled=True,
                              learn_rate=0.1, beta=1.0, lambda_=0.01, using_adagrad=True, user_factor=True, batch_users=B1)
    model = cdae_amd.MultiCDAE(cfg, devices=devices, exchange_every=0)
    combine = cdae_amd.COMBINE_GLOBAL_ACC if args.combine == "global-acc" else cdae_amd.COMBINE_SUM
    model.set_schedule(period=0, combine=combine, sync_batch_users=sb, relay_epochs=1.0)
    model.reset(data, seed=args.seed)
    t_r = time.perf_counter()
    with WATCHDOG.stage("relayed epoch (single-GPU schedule, shard by shard)", WD_SYNC_S):
        model.train_one_iteration(args.seed, 0)                       # epoch 0: the relay (warm-up; untimed)
    relay_s = time.perf_counter() - t_r
    spe = max(1, model.steps_per_epoch)

    def run(first, count):
        """exchanged steps first .. first + count - 1, counted from the start of epoch 1; returns users trained"""
        users, i, end = 0, first, first + count
        while i < end:
            ep, t = 1 + i // spe, i % spe
            n = min(end - i, spe - t)
            users += model.train_steps(args.seed, ep, t, t + n).users
            i += n
        return users

    with WATCHDOG.stage("warm-up steps", WD_SYNC_S):
        run(0, args.warmup)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with WATCHDOG.stage("timed steps", WD_SYNC_S):
        users = run(args.warmup, args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    n_gpus = len(set(devices))
    Kp = 64 * (1 if K <= 64 else 2 if K <= 128 else 4 if K <= 256 else 8)
    users_step = users / max(1, args.steps)
    ex_shard_step = data.nnz_train * 6.0 / data.num_users * users_step / n_shards
    comp = compulsory_decode_bytes(K, Kp, data.num_items, ex_shard_step, users_step / n_shards)
    achieved = comp / (elapsed / args.steps) / 1e9
    value = users / elapsed
    single = (side.get("single_gpu") or {}).get("users_per_s")
    out = {"metric": ("users/sec (whole node) K=200 ML-10M-shape; Recall@10 parity" if (args.shape == "ml10m" and K == 200)
                      else f"users/sec (whole node) K={K} {args.shape}-shape"),
           "value": value, "unit": "users/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{args.shape}-shape synthetic {data.num_users}x{data.num_items} (ONE data set, users sharded over {n_shards} "
                                  f"{'logical shards of one GPU' if n_gpus == 1 and n_shards > 1 else 'GPUs'} by interactions), nnz_train={data.nnz_train}, "
                                  f"K={K}, num_neg=5, CE loss, AdaGrad, q=0.5 scaled",
                      "batch_users": sb, "global_batch": int(round(users_step)), "parallelism": f"dp{n_shards} (user shards, one process, a host thread per GPU)",
                      "exchange": f"synchronous: after every step ONE all-reduce(sum) of the accumulated delta of the shared block "
                                  f"[W | W_ag | b' | b'_ag | b | b_ag] (library-owned RCCL communicators), {args.combine} combine; Wu stays on its shard",
                      "schedule": {"relay_epochs": 1.0, "relay_epoch_seconds_untimed": relay_s, "sync_batch_users": sb, "steps_per_epoch": spe,
                                   "note": "the relayed epoch is the single-GPU schedule handed from GPU to GPU (one GPU's time, once per training run: 1 of "
                                           "apps/yelp's 50 epochs); the timed steps are the exchanged schedule of every later epoch"},
                      "accuracy": ("relay 1.0 + %d users per GPU and step + %s: mean-over-seeds Recall@10 within +-0.002 of the sequential reference from epoch 2 on at "
                                   "8 x 64 users per step (ML-10M shape: <= 0.0014; Netflix shape: 0.0020, at the edge), single seeds within 0.009 / 0.005 — "
                                   "the schedule's OWN bounds, wider than the single GPU's (tests/test_gpu_accuracy.py::test_relay_then_exchange_schedule_on_eight_"
                                   "shards_at_ml10m_shape, tests/test_gpu_netflix.py::test_netflix_relay_then_exchange_schedule_on_eight_shards; DESIGN.md §7)"
                                   % (sb, args.combine)) if sb <= 64 and n_shards <= 8 else
                                  "OUTSIDE the measured envelope (more than 64 users per GPU and step, or more than 8 shards): throughput only",
                      "same_node_alternatives": side,
                      "vs_single_gpu": (value / single) if single else None},
           "roofline": {"bound": "hbm", "kernel": "whole exchanged step per GPU (decode + gather launch of the shard's 64 users, stage, all-reduce, merge)",
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                        "frac_definition": "one shard's compulsory decode bytes / WHOLE step time / HBM peak: the step is bound by the all-reduce of the whole "
                                           "shared block and the stage / merge passes over it (DESIGN.md §8), not by the decode"}}
    model.close()
    if dist is not None:
        dist.destroy_process_group()
    print(json.dumps(out), flush=True)


FULL_OUTPUT_CERTIFIED_BLOCK = 512     # largest block size the driver-run full-output accuracy test holds (tests/test_gpu_accuracy.py)


def full_output_accuracy(B, num_users, shape=None, K=None):
    """what a full-output line may claim (tools/accuracy_envelope.py --full-output, three shapes x four seeds)"""
    head = (f"full-output BLOCK schedule: one summed AdaGrad step per decoder row per block of {B} users - a different (faster) optimizer of the same "
            "objective than its B = 1 limit, the reference loop cdae.hpp:225-293 fed every unrated item; trajectory-equal to that loop at NO block size "
            "above 1 (DESIGN.md §5c); ")
    measured = {   # builder-run envelopes, four seeds each (profiles/r04_full_output_envelope_*.txt): epochs to the loop's best Recall@10
        ("ml10m", 200, 2048): "at this shape and block size (tests/test_gpu_accuracy.py::test_ml10m_shape_bench_block_reaches_the_literal_loops_quality, driver-run, two "
                              "seeds; four in DESIGN.md §5c): reaches the loop's 25-epoch best Recall@10 (0.161) within 15-16 epochs (0.11 s of training) and is "
                              "above it through epoch 25 (0.171); smaller blocks get there sooner (512 users: 5-6 epochs) but over-train afterwards (Recall@10 "
                              "declines from epoch ~10 on)",
        ("ml10m", 200, 512): "measured at this shape and block size: reaches the loop's 25-epoch best Recall@10 (0.161) within 5-6 epochs, peaks at 0.167 around "
                             "epoch 10 and then over-trains (0.152 at epoch 25)",
    }
    note = measured.get((shape, K, B))
    if B <= FULL_OUTPUT_CERTIFIED_BLOCK:
        txt = head + ("certified ONE-SIDEDLY as a mean over four seeds at Yelp shape K=50 (tests/test_gpu_accuracy.py::"
                      "test_full_output_block_schedule_reaches_the_literal_loops_quality): Recall@10 reaches the loop's 30-epoch best within 4 / 5 / 7 / 11 / 16 / 27 "
                      "epochs at 16 / 32 / 64 / 128 / 256 / 512 users per block - 0.12 / 0.079 / 0.054 / 0.044 / 0.035 / 0.036 s of training on one MI355X - and "
                      "stays above it there")
    else:
        txt = head + (f"this block size is ABOVE the ones a driver-run test certifies (<= {FULL_OUTPUT_CERTIFIED_BLOCK} at Yelp shape): what decides is block steps per "
                      f"epoch = users / block ({num_users / B:.0f} on this data set): ~400-1000 block steps reach the loop's best at every measured shape")
    return txt + ("; " + note if note else "")


def measured_traffic(shape, K, B):
    """HBM bytes per decode launch from the committed rocprofv3 PMC passes (profiles/*_decode_traffic.json), or None
    when no pass was taken for this exact workload.  bench.py cannot collect PMC counters on itself."""
    import glob
    best, source = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_decode_traffic.json"))):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        if t.get("shape") == shape and t.get("num_dim") == K and t.get("batch_users") == B:
            best, source = t.get("traffic_bytes_per_launch"), os.path.relpath(path, ROOT) + " (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; not measured in this run)"
    return best, source


def measured_full_traffic(shape, K, B):
    """HBM bytes per launch of the K > 256 full-output step's dominant launch (gemm3_rows_fused_kernel) from the committed rocprofv3
    PMC passes (profiles/*_full_traffic.json: FETCH_SIZE x 2 per the guide's gfx950 note + WRITE_SIZE), or None."""
    import glob
    best, source = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_full_traffic.json"))):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        if t.get("shape") == shape and t.get("num_dim") == K and t.get("batch_users") == B:
            best, source = t.get("rows_fused_bytes_per_launch"), os.path.relpath(path, ROOT) + " (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; not measured in this run)"
    return best, source


def cpu_baseline(data, cfg, args):
    """The reference's per-user sequential fp64 algorithm (oracle, literal schedule), ONE thread — the
    reference training loop is single-threaded (cdae.hpp:136-146) — on a bounded sample."""
    import oracle as orc
    n = min(args.cpu_users, data.num_users)
    ocfg = orc.OracleConfig(num_dim=cfg.num_dim, loss_type=cfg.lt, num_neg=cfg.num_neg,
                            corruption_ratio=cfg.corruption_ratio, scaled=cfg.scaled, learn_rate=cfg.learn_rate,
                            beta=cfg.beta, lambda_=cfg.lambda_)
    o = orc.Oracle(ocfg, data.num_users, data.num_items, data.train_ptr, data.train_col)
    o.init_params(args.seed)
    o.train_literal(args.seed, 0, 0, min(200, n))     # warm caches
    t0 = time.perf_counter()
    o.train_literal(args.seed, 0, 0, n)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "users/s", "cores": 1, "kind": "port",
            "sample": f"first {n} users of the same shard, one pass of the literal cdae.hpp:136-358 restatement "
                      f"(fp64, g++ -O3), {dt:.1f} s; host has {os.cpu_count()} logical cores"}


if __name__ == "__main__":
    main()
