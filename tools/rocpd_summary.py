#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into the text files kept under profiles/.

  python tools/rocpd_summary.py stats  <results.db>            # --kernel-trace --stats summary
  python tools/rocpd_summary.py pmc    <results.db> COUNTER    # per-kernel mean of one PMC counter
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
    else:
        pmc(sys.argv[2], sys.argv[3])
