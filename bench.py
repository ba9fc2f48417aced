#!/usr/bin/env python
"""bench.py — users/sec of the CDAE training hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] — ML-10M-shape synthetic (70K users x 10.6K items,
~10M interactions, 80/20 per-user split), K=200, num_neg=5, sigmoid hidden, cross-entropy loss, AdaGrad.
A step = one pass of the hot path (sample -> sort -> encode -> row-major decode -> hidden -> input rows)
over one batch of `batch_users` users, cycling through the shard; with N > 1 every rank (one process per
GPU, launched by torch.distributed.run) trains its OWN ML-10M-shaped shard (weak scaling: per-GPU work
is fixed) and each step ends with one RCCL all-reduce of the shared-parameter deltas.
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
# tests/test_gpu_accuracy.py trains at THIS value and asserts |dRecall@10| <= 0.002 against the literal-schedule fixtures
# at every epoch for every seed (DESIGN.md §2 has the sweep)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=274)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-users", type=int, default=int(os.environ.get("CDAE_BATCH_USERS", DEFAULT_BATCH_USERS)),
                    help="users per parameter snapshot; the default is the largest value whose Recall@10 stays within +-0.002 of "
                         "the sequential reference at every epoch for every fixture seed (tests/test_gpu_accuracy.py)")
    ap.add_argument("--shape", default="ml10m")
    ap.add_argument("--num-dim", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-users", type=int, default=40000, help="users in the timed CPU-baseline sample (~20 s)")
    ap.add_argument("--seed", type=int, default=20141119)
    ap.add_argument("--profile-every", type=int, default=8, help="HIP-event kernel timing on every n-th batch (0 = off)")
    ap.add_argument("--full-output", action="store_true", help="BASELINE configs[1]/[4]: every unrated item is a negative; "
                    "dense decode on the bf16 MFMA cores (roofline bound: mfma)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) in production; gloo only to smoke-test the N>1 "
                    "code path on a single-GPU box together with --share-device")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (functional test only)")
    ap.add_argument("--exchange-every", type=int, default=-1, help="N > 1: batches between exchanges of the shared-parameter "
                    "deltas (pipelined: the all-reduce overlaps the next period).  -1 (default): chosen at start-up so that one "
                    "period of training covers a measured all-reduce; 0: synchronous exchange after every batch")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the CDAE hot path has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import cdae_amd
    from cdae_amd import synth
    from cdae_amd.distributed import DeltaExchange, PipelinedDeltaExchange

    # every rank owns one ML-10M-shaped shard of users over the same item space
    data = synth.generate_shape(args.shape, seed=args.seed + 7919 * rank)
    K, B = args.num_dim, min(args.batch_users, data.num_users)
    cfg = cdae_amd.CDAEConfig(num_dim=K, lt=cdae_amd.CROSS_ENTROPY, num_neg=5, num_corruptions=1,
                              corruption_ratio=0.5, scaled=True, learn_rate=0.1, beta=1.0, lambda_=0.01,
                              using_adagrad=True, user_factor=True, batch_users=B, full_output=args.full_output)
    model = cdae_amd.CDAE(cfg, device=local_rank)
    model.set_interactions(data.num_users, data.num_items, data.train_ptr, data.train_col,
                           user_id_offset=rank * data.num_users)
    model.init_params(args.seed)         # identical shared parameters on every rank; Wu differs but is private
    # The process group is created AFTER the library handle: with an RCCL communicator (and its streams) in place first, the
    # handle's streams are assigned hardware queues that make the overlapped exchange several times slower (measured with
    # a one-rank group: 0.42 vs 0.18 ms per step).
    dist = None
    force_dist = bool(os.environ.get("CDAE_BENCH_FORCE_DIST"))   # developer aid: exercise the RCCL path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    exch = pipe = None
    if (world > 1 or force_dist) and args.exchange_every != 0:
        pipe = PipelinedDeltaExchange(model, dist, world, period=max(1, args.exchange_every))
    elif world > 1 or force_dist:
        exch = DeltaExchange(model, dist, world)

    n_batches = (data.num_users + B - 1) // B

    def batch_of(i):
        b = i % n_batches
        return i // n_batches, b * B, min(data.num_users, (b + 1) * B)

    KEYS = ("users", "examples", "batches", "ms_sample", "ms_sort", "ms_encode", "ms_decode", "ms_hidden", "ms_input", "launches_decode")
    acc = {k: 0 for k in KEYS}

    def add(st):
        for k in KEYS:
            acc[k] += getattr(st, k)

    def step(i):
        ep, u0, u1 = batch_of(i)
        if exch:
            exch.begin()
        # steps queue asynchronously on the library's stream; the next batch is sampled and sorted on the side
        # stream while this one trains (and, with N > 1, while its deltas are all-reduced)
        model.enqueue_users(args.seed, ep, u0, u1)
        nep, n0, n1 = batch_of(i + 1)
        model.prefetch_users(args.seed, nep, n0, n1)
        if exch:
            exch.finish()
        if pipe:
            pipe.after_batch()

    def sync():
        if dist is not None:
            dist.barrier()
        model.synchronize()
        torch.cuda.synchronize()

    exchange_note = None
    if pipe and args.exchange_every < 0:
        # auto period: time a few batches without exchange and a few idle all-reduces of the real buffer, before anything
        # that counts has been staged
        pipe.period = 1 << 30
        sync()
        n_cal = max(3, min(args.warmup, 10))        # calibration batches (set-up, before the W warm-up steps)
        step(0)                                      # first batch: cold pipeline, not timed
        sync()
        tw = time.perf_counter()
        for i in range(1, 1 + n_cal):
            step(i)
        sync()
        t_step = (time.perf_counter() - tw) / n_cal
        pipe.flush()                      # exchange what those batches did, so the replicas agree again
        args.exchange_every, t_ar = pipe.choose_period(t_step, lo=2)       # a boundary costs ~40 us of stream time: never every batch
        exchange_note = f"period chosen at start-up: all-reduce {t_ar * 1e6:.0f} us vs {t_step * 1e6:.0f} us per batch"
    for i in range(args.warmup):
        step(i)
    model.collect_stats()
    acc = {k: 0 for k in KEYS}
    # HIP events on the library's own streams around each kernel family of every `profile_every`-th batch of the timed
    # region (event records cost ~3 us of stream time each: 42 us per step if every batch carries its 14)
    model.set_profiling(args.profile_every)
    sync()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    if pipe:
        pipe.flush()                      # the last period's deltas are reduced and merged inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    add(model.collect_stats())
    model.set_profiling(False)

    users_total = float(acc["users"])
    if dist is not None:
        t = torch.tensor([elapsed, users_total], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, users_total = float(tmax[0]), float(t[1])

    if dist is not None:
        # RCCL writes a version banner to the C stdout buffer of every rank: push it out now, on all ranks, so that
        # rank 0's JSON line is the last thing the job prints
        import ctypes
        ctypes.CDLL(None).fflush(None)
        dist.barrier()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    n_u = data.nnz_train / data.num_users
    n_in = n_u * (1.0 - cfg.corruption_ratio)
    a_user = algorithmic_bytes_per_user(K, n_u, n_in, cfg.num_neg)
    value = users_total / elapsed
    # roofline of the dominant kernel (decode_rows_kernel), rank 0's launches
    ex_per_launch = acc["examples"] / max(1, acc["batches"])
    ms_per_launch = acc["ms_decode"] / max(1, acc["launches_decode"])
    alg_bytes_launch = decode_bytes_per_example(K) * ex_per_launch
    achieved = alg_bytes_launch / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
    traffic = measured_traffic(args.shape, K, B)
    if args.full_output:
        # dominant kernels: the three bf16 MFMA contractions, 6 K I flop per user, timed as one family
        MFMA_PEAK_TFLOPS = 2500.0       # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
        flops_launch = 6.0 * K * data.num_items * (acc["users"] / max(1, acc["batches"]))
        achieved_tf = flops_launch / (ms_per_launch * 1e-3) / 1e12 if ms_per_launch > 0 else 0.0
        roofline = {"bound": "mfma", "kernel": "full_decode_fused_kernel + gemm_nt_bf16_kernel (+ bf16 operand copies, rated-items bitmap)",
                    "achieved": achieved_tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tf / MFMA_PEAK_TFLOPS,
                    "traffic": None, "algorithmic_flops_per_launch": flops_launch, "avg_launch_ms": ms_per_launch}
        workload = (f"{args.shape}-shape synthetic {data.num_users}x{data.num_items} per GPU, nnz_train={data.nnz_train}, K={K}, "
                    f"FULL-OUTPUT decode (every unrated item a negative), CE loss, AdaGrad, q=0.5 scaled")
    else:
        roofline = {"bound": "hbm", "kernel": "decode_hybrid_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg_bytes_launch, "avg_launch_ms": ms_per_launch,
                    "whole_step_fraction_of_hbm_roof": value / args.gpus * a_user / 1e9 / HBM_PEAK_GBS}
        workload = (f"{args.shape}-shape synthetic {data.num_users}x{data.num_items} per GPU, nnz_train={data.nnz_train}, K={K}, "
                    f"num_neg=5, CE loss, AdaGrad, q=0.5 scaled")
    out = {
        # BASELINE.json's metric on its own workload; other shapes / K (developer runs) are named as what they are
        "metric": ("users/sec (whole node) K=200 ML-10M-shape; Recall@10 parity" if (args.shape == "ml10m" and args.num_dim == 200)
                   else f"users/sec (whole node) K={args.num_dim} {args.shape}-shape"),
        "value": value, "unit": "users/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.full_output else "f32", "data": "synthetic",
        "config": {"workload": workload, "batch_users": B, "global_batch": B * args.gpus, "parallelism": f"dp{args.gpus}",
                   "exchange": ("none" if (pipe is None and exch is None) else
                                f"pipelined all-reduce(sum) of shared-parameter deltas every {args.exchange_every} batches, merged one period late"
                                + (f" ({exchange_note})" if exchange_note else "")
                                if args.exchange_every > 0 else "synchronous all-reduce(sum) of shared-parameter deltas every batch")},
        "roofline": roofline,
        "kernel_ms_per_step": {k[3:]: acc[k] / max(1, acc["launches_decode"]) for k in acc if k.startswith("ms_")},
        "profiled_steps": int(acc["launches_decode"]),
    }
    if not args.no_cpu_baseline and args.gpus == 1:          # reported at N = 1 only (rank 0's host cores)
        out["cpu_baseline"] = cpu_baseline(data, cfg, args)
    if dist is not None:
        dist.destroy_process_group()
        import ctypes
        ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


def measured_traffic(shape, K, B):
    """HBM bytes per decode launch from the committed rocprofv3 PMC passes (profiles/*_decode_traffic.json), or None
    when no pass was taken for this exact workload.  bench.py cannot collect PMC counters on itself."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_decode_traffic.json"))):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        if t.get("shape") == shape and t.get("num_dim") == K and t.get("batch_users") == B:
            best = t.get("traffic_bytes_per_launch")
    return best


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
