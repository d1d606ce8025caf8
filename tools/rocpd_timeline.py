#!/usr/bin/env python
"""Per-launch timeline of the kernels whose name contains SUBSTR out of rocprofv3 rocpd databases (`--kernel-trace`): start
(us since the first listed launch), duration, queue / stream, in start order -- which bulk launches ran beside which side-stream
launch and what that did to them (the summary's min / avg / max does not say).
usage: python tools/rocpd_timeline.py <results.db | dir> SUBSTR[,SUBSTR..] [--only SUBSTR] [--limit N]"""
import glob
import os
import sqlite3
import sys


def main(argv):
    paths, subs, only, limit = [], None, None, 400
    it = iter(argv)
    for a in it:
        if a == "--only":
            only = next(it)
        elif a == "--limit":
            limit = int(next(it))
        elif os.path.isdir(a):
            paths += sorted(glob.glob(os.path.join(a, "**", "*.db"), recursive=True))
        elif os.path.exists(a):
            paths.append(a)
        else:
            subs = a.split(",")
    for p in paths:
        db = sqlite3.connect(p)
        try:
            cur = db.execute("select * from kernels")
            cols = [c[0] for c in cur.description]
            rows = [dict(zip(cols, r)) for r in cur.fetchall()]
        except sqlite3.Error as e:
            print("#", p, e)
            continue
        finally:
            db.close()
        if only and not any(only in r["name"] for r in rows):
            continue
        rows = [r for r in rows if any(s in r["name"] for s in subs)]
        if not rows:
            continue
        rows.sort(key=lambda r: r["start"])
        t0 = rows[0]["start"]
        qk = next((k for k in ("queue_id", "queue", "stream_id", "stream") if k in cols), None)
        print("# %s: %d launches; columns: %s" % (os.path.basename(p), len(rows), ",".join(cols)))
        print("%10s %9s %6s  %s" % ("start_us", "dur_us", qk or "-", "kernel"))
        for r in rows[:limit]:
            print("%10.1f %9.2f %6s  %s" % ((r["start"] - t0) / 1e3, (r["end"] - r["start"]) / 1e3, r.get(qk, "-") if qk else "-",
                                          r["name"].split("(")[0][-40:]))


if __name__ == "__main__":
    main(sys.argv[1:])
