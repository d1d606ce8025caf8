#!/bin/bash
mkdir -p gpurun_out
for s in 8 12 16 24 48; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra --fused 0 --straggler-ticks $s > gpurun_out/r2g_s${s}.json 2> gpurun_out/r2g_s${s}.err
    python - "straggler_ticks=$s" gpurun_out/r2g_s${s}.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-500:])
PY
done 2>&1 | tee gpurun_out/r2g_sweep.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2g_prof -- python $R/bench.py --no-cpu --no-rs --steps 12 --warmup 4 > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $(find gpurun_out/r2g_prof -name "*.db" | head -1) 2>&1 | grep -v "at::native" | head -30 | cut -c1-150 | tee gpurun_out/r2g_kernel_stats.txt
