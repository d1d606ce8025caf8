#!/usr/bin/env python
"""Summarise rocprofv3 rocpd SQLite databases (`--kernel-trace --stats`) into a per-kernel text table: calls, total /
average / min / max duration (us), share.  A profiled command that starts child processes (bench.py runs every secondary
leg in its own) leaves ONE database per process: pass a directory (searched recursively) or several files and the table
covers all of them; `--only SUBSTR` keeps the databases that hold a kernel whose name contains SUBSTR (e.g. the process of
the headline region: `--only mp_quorum_tally`).
usage: python tools/rocpd_summary.py <results.db | dir> [...] [--only SUBSTR] [> profiles/NAME_kernel_stats.txt]"""
import glob
import os
import sqlite3
import sys


def kernel_rows(path):
    db = sqlite3.connect(path)
    try:
        return db.execute("select name, duration, vgpr_count, sgpr_count, lds_size, scratch_size from kernels").fetchall()
    except sqlite3.Error:
        return []
    finally:
        db.close()


def main(argv):
    only, paths = None, []
    it = iter(argv)
    for a in it:
        if a == "--only":
            only = next(it)
        elif os.path.isdir(a):
            paths += sorted(glob.glob(os.path.join(a, "**", "*.db"), recursive=True))
        else:
            paths.append(a)
    acc, used = {}, []
    for p in paths:
        rows = kernel_rows(p)
        if only is not None and not any(only in r[0] for r in rows):
            continue
        used.append((p, len(rows)))
        for n, d, vg, sg, lds, scr in rows:
            e = acc.setdefault(n, [0, 0, 1 << 62, 0, vg, sg, lds, scr])
            e[0] += 1; e[1] += d; e[2] = min(e[2], d); e[3] = max(e[3], d)
    print("# %d database(s): %s" % (len(used), ", ".join("%s (%d dispatches)" % (os.path.basename(p), n) for p, n in used)))
    total = sum(e[1] for e in acc.values()) or 1
    print("%-72s %7s %12s %10s %10s %10s %6s %5s %5s %6s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us",
                                                                 "max_us", "pct", "vgpr", "sgpr", "lds", "scratch"))
    for n, (c, s, mn, mx, vg, sg, lds, scr) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5s %5s %6s %7s" %
              (n[:72], c, s / 1e3, s / c / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / total, vg, sg, lds, scr))


if __name__ == "__main__":
    main(sys.argv[1:])
