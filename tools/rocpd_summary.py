#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite result: per-kernel calls / total / avg / min / max
(the --stats table) and, if present, PMC counter sums per kernel.  Usage:
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# kernel-trace stats from {path}")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6}  name")
    for name, calls, tot, avg, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"{calls:>7} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100 * tot / total:>6.2f}  {short}")
    try:
        pm = cur.execute("select k.name, p.counter_name, sum(p.value), count(*) from pmc_events p "
                         "join kernels k on p.dispatch_id = k.dispatch_id group by 1,2 order by 1,2").fetchall()
    except sqlite3.Error:
        try:
            ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            kn = "kernel_name" if "kernel_name" in ccols else "name"
            pm = cur.execute(f"select {kn}, counter_name, sum(value), count(*) from counters_collection "
                             f"group by 1,2 order by 1,2").fetchall()
        except sqlite3.Error:
            pm = []
    if pm:
        print("\n# PMC counters (sum over dispatches, per-dispatch mean)")
        for name, cname, val, cnt in pm:
            short = name if len(name) < 80 else name[:77] + "..."
            print(f"{cname:<28} sum={val:>18.1f} n={cnt:>6} mean={val / cnt:>16.1f}  {short}")


if __name__ == "__main__":
    main(sys.argv[1])
