#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into the text files kept under profiles/.

  python tools/rocpd_summary.py stats  <results.db>            # --kernel-trace --stats summary
  python tools/rocpd_summary.py pmc    <results.db> COUNTER    # per-kernel mean of one PMC counter
  python tools/rocpd_summary.py timeline <results.db> [N]      # the last N kernel dispatches: start / end (us from the first), queue, name
"""
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name if len(name) < 110 else name[:107] + "..."


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name "
                       "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':110s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'scr':>4s}")
    for n, c, tot, avg, mn, mx, vg, sg, lds, scr in rows:
        print(f"{short(n):110s} {c:6d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:6.2f} {vg or 0:5d} {sg or 0:5d} {lds or 0:6d} {scr or 0:4d}")


def timeline(db, n):
    cur = sqlite3.connect(db).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute(f"select name, start, end, {qcol or 0} from kernels order by start desc limit ?", (n,)).fetchall()[::-1]
    t0 = rows[0][1]
    prev_end = {}
    print(f"{'start_us':>10s} {'end_us':>10s} {'dur_us':>8s} {'gap_same_queue':>14s} {'queue':>6s}  kernel")
    for name, a, b, q in rows:
        gap = (a - prev_end[q]) / 1e3 if q in prev_end else float("nan")
        prev_end[q] = b
        print(f"{(a - t0) / 1e3:10.2f} {(b - t0) / 1e3:10.2f} {(b - a) / 1e3:8.2f} {gap:14.2f} {q!s:>6s}  {short(name)[:90]}")


def pmc(db, counter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    name_col = "counter_name" if "counter_name" in cols else "name"
    val_col = "value" if "value" in cols else "counter_value"
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = cur.execute(f"select {kcol}, count(*), avg({val_col}), min({val_col}), max({val_col}) from counters_collection "
                       f"where {name_col} = ? group by {kcol} order by avg({val_col}) desc", (counter,)).fetchall()
    print(f"# counter {counter}: per-dispatch values (raw units as reported by rocprofv3)")
    print(f"{'kernel':110s} {'dispatches':>10s} {'mean':>16s} {'min':>16s} {'max':>16s}")
    for n, c, avg, mn, mx in rows:
        print(f"{short(n):110s} {c:10d} {avg:16.1f} {mn:16.1f} {mx:16.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
    else:
        pmc(sys.argv[2], sys.argv[3])
