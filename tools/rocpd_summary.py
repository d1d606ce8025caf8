#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (`--kernel-trace --stats`) into a
per-kernel text table: calls, total / average / min / max duration (us), share.
usage: python tools/rocpd_summary.py <results.db> [> profiles/NAME_kernel_stats.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-72s %7s %12s %10s %10s %10s %6s %5s %5s %6s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us",
                                                                 "max_us", "pct", "vgpr", "sgpr", "lds", "scratch"))
    for n, c, s, a, mn, mx, vg, sg, lds, scr in rows:
        print("%-72s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5s %5s %6s %7s" %
              (n[:72], c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / total, vg, sg, lds, scr))
    try:
        pmc = db.execute("select * from counters_collection limit 1").fetchall()
        if pmc:
            print("\n# PMC counters (avg per dispatch)")
            cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
            ni, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else None, None, None
            for cand in ("counter_name", "name"):
                if cand in cols:
                    ci = cols.index(cand)
            for cand in ("value", "counter_value"):
                if cand in cols:
                    vi = cols.index(cand)
            if None not in (ni, ci, vi):
                q = "select %s, %s, avg(%s), count(*) from counters_collection group by 1, 2 order by 1, 2" % (
                    cols[ni], cols[ci], cols[vi])
                for k, cn, v, n in db.execute(q):
                    print("%-60s %-24s %18.1f  (n=%d)" % (k[:60], cn, v, n))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
