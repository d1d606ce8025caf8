#!/bin/bash
# r2z6: straggler ttl x batch size at the driver's flags and at the defaults
mkdir -p gpurun_out
for a in "--steps 20 --warmup 5 --batch 8 --straggler-ticks 2" "--steps 20 --warmup 5 --batch 8 --straggler-ticks 4" "--steps 20 --warmup 5 --batch 10 --straggler-ticks 4" "--steps 20 --warmup 5 --batch 5 --straggler-ticks 4" \
         "--batch 8 --straggler-ticks 2" "--batch 8 --straggler-ticks 4" "--batch 12 --straggler-ticks 4" "--batch 6 --straggler-ticks 4"; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2z6.json 2> gpurun_out/r2z6.err
    python - "args=[$a]" gpurun_out/r2z6.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
done 2>&1 | tee gpurun_out/r2z6_ttl_batch.log
