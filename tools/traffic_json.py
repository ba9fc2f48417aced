#!/usr/bin/env python
"""profiles/<round>_decode_traffic.json and profiles/<round>_full_traffic.json from the round's rocprofv3 PMC summaries
(tools/rocpd_summary.py pmc output, one file per counter): HBM bytes per launch of the kernel a bench line's roofline is about —
bench.py reads them back as `roofline.traffic` (a process cannot collect PMC counters on itself).

    python tools/traffic_json.py decode r04 profiles/r04_bench_pmc_fetch_size.txt profiles/r04_bench_pmc_write_size.txt \
        profiles/r04_bench_kernel_stats.txt --shape ml10m --num-dim 200 --batch-users 256
    python tools/traffic_json.py full r04 profiles/r04_full_output_cfg5_pmc.txt - profiles/r04_full_output_cfg5_kernel_stats.txt \
        --shape cfg5_items --num-dim 512 --batch-users 1024

rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE is doubled per the gfx950 note of /opt/skills/guides/MI355X_MICROARCH.md's
HBM section, WRITE_SIZE is taken as reported — the same corrections as rounds 1-3 (whose JSON files were typed in by hand)."""
import argparse
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_of(path, kernel, counter=None):
    """mean per-dispatch value of `kernel` in a rocpd_summary pmc file (optionally inside the block of `counter`)"""
    inside = counter is None
    for line in open(path):
        if line.startswith("# counter"):
            inside = counter is None or f"counter {counter}:" in line
            continue
        if inside and kernel in line:
            nums = re.findall(r"[-+]?\d+\.\d+|\d+", line[100:])
            return float(nums[1])            # dispatches, mean, min, max
    raise SystemExit(f"{kernel} not found in {path}" + (f" ({counter})" if counter else ""))


def avg_us(stats_path, kernel):
    for line in open(stats_path):
        if kernel in line:
            return float(line[110:].split()[2])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=["decode", "full"])
    ap.add_argument("round")
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("stats")
    ap.add_argument("--shape", required=True)
    ap.add_argument("--num-dim", type=int, required=True)
    ap.add_argument("--batch-users", type=int, required=True)
    ap.add_argument("--kernel", default="decode_gather_kernel", help="decode: the launch the roofline is about (decode_hybrid_kernel through round 5)")
    a = ap.parse_args()
    if a.kind == "decode":
        k = a.kernel
        f, w = mean_of(a.fetch, k), mean_of(a.write, k)
        out = {"_comment": f"HBM-side traffic of cdae::{k} per launch: two separate rocprofv3 PMC passes (--kernel-trace --pmc FETCH_SIZE / WRITE_SIZE) of "
                           f"`python bench.py --no-cpu-baseline --steps 120 --warmup 20` (tools/profile_round.sh {a.round}): {os.path.relpath(a.fetch, ROOT)}, "
                           f"{os.path.relpath(a.write, ROOT)}.  FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note, WRITE_SIZE as reported (KiB).  "
                           f"rocprofv3 --kernel-trace average duration of the kernel in the same session: {avg_us(a.stats, k)} us.",
               "kernel": k, "batch_users": a.batch_users, "shape": a.shape, "num_dim": a.num_dim,
               "fetch_size_kb_reported": f, "write_size_kb_reported": w, "traffic_bytes_per_launch": int((2.0 * f + w) * 1024)}
        name = f"{a.round}_decode_traffic.json"
    else:
        recs = {}
        for k in ("gemm3_rows_fused_kernel", "gemm1_loss_duo_kernel", "gemm_tn_bf16_kernel"):
            f, w = mean_of(a.fetch, k, "FETCH_SIZE"), mean_of(a.fetch, k, "WRITE_SIZE")
            recs[k] = {"fetch_size_kb_reported": f, "write_size_kb_reported": w, "bytes_per_launch": int((2.0 * f + w) * 1024), "avg_us": avg_us(a.stats, k)}
        out = {"_comment": f"HBM-side traffic per launch of the K > 256 full-output step's three big launches: separate rocprofv3 PMC passes of `python bench.py "
                           f"--no-cpu-baseline --full-output --shape cfg5_items --num-dim 512 --batch-users 1024` (tools/profile_full_output.sh {a.round}): "
                           f"{os.path.relpath(a.fetch, ROOT)}.  FETCH_SIZE doubled per the guide's gfx950 note, WRITE_SIZE as reported (KiB).",
               "batch_users": a.batch_users, "shape": a.shape, "num_dim": a.num_dim, "launches": recs,
               "rows_fused_bytes_per_launch": recs["gemm3_rows_fused_kernel"]["bytes_per_launch"]}
        name = f"{a.round}_full_traffic.json"
    path = os.path.join(ROOT, "profiles", name)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, {k: v for k, v in out.items() if k != "_comment"})


if __name__ == "__main__":
    main()
