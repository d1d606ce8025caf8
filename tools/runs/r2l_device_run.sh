#!/bin/bash
# the -m gpu suite with the Heartbeater / string-KV kernels, then a kernel trace of the DEFAULT bench line (side stream on)
# next to its own event-pair figures
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r2l_gputests.log; tail -2 gpurun_out/r2l_gputests.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2l_prof -- python $R/bench.py --no-cpu --no-rs --no-extra > $R/gpurun_out/r2l_bench_prof.json 2> $R/gpurun_out/r2l_bench_prof.err
cd $R
for f in $(find gpurun_out/r2l_prof -name "*.db"); do python tools/rocpd_summary.py $f | grep -v "at::native" | head -12 | cut -c1-150; done | tee gpurun_out/r2l_kernel_stats.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2l_bench_prof.json").read().strip().splitlines()[-1])
print("bench under rocprof: value %.3e ms/tick %.4f" % (d["value"], d["ms_per_step"]), {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
PY
