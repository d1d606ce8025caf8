#!/bin/bash
# r2z5: what slows the bulk kernels when 1 % of the groups change leader: all timeouts inside the warm-up (--timeout-span 10), so
# the timed region has no leader change in flight, only groups whose leader moved -- back with the bulk (ttl 8) or kept on the list (ttl 200)
mkdir -p gpurun_out
for a in "--timeout-span 10 --batch 8" "--timeout-span 10 --batch 8 --straggler-ticks 200" "--timeout-span 10" "--timeout-span 10 --straggler-ticks 0" "--timeouts 0 --batch 8"; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2z5.json 2> gpurun_out/r2z5.err
    python - "args=[$a]" gpurun_out/r2z5.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
done 2>&1 | tee gpurun_out/r2z5_moved.log
