#!/bin/bash
mkdir -p gpurun_out
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/r3l_prof -- python $R/bench.py --layout spread --spread-ranks 4 --steps 24 --warmup 6 > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/r3l_prof > gpurun_out/r3l_kernel_stats_spread4.txt 2>&1
python - <<'P'
import sqlite3, glob
db = sqlite3.connect(glob.glob("gpurun_out/r3l_prof/**/*.db", recursive=True)[0])
rows = db.execute("select start, end from kernels order by start").fetchall()
# busy time (union of intervals) over the last 60% of the run
t0 = rows[int(len(rows)*0.4)][0]; t1 = rows[-1][1]
busy = 0; cur_s = cur_e = None
for s, e in rows:
    if s < t0: continue
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window ms", (t1 - t0) / 1e6, "kernel-busy ms", busy / 1e6, "kernels in window", sum(1 for s, e in rows if s >= t0))
try:
    n = db.execute("select count(*), sum(end-start) from memory_copies").fetchall()
    print("memory copies", n)
except Exception as ex:
    print("no memcpy table", ex)
P
rm -rf gpurun_out/r3l_prof
head -22 gpurun_out/r3l_kernel_stats_spread4.txt | cut -c1-170
