#!/bin/bash
# r2z3: how long a group stays on the straggler list (a group whose leader moved makes its wavefront of the bulk kernels run
# both the leader's and the follower's path) x ticks per batch
mkdir -p gpurun_out
for a in "--batch 8 --straggler-ticks 32" "--batch 8 --straggler-ticks 200" "--batch 16 --straggler-ticks 200" "--straggler-ticks 200" \
         "--steps 20 --warmup 5 --batch 8 --straggler-ticks 32" "--steps 20 --warmup 5 --batch 8 --straggler-ticks 200" \
         "--steps 20 --warmup 5 --batch 16 --straggler-ticks 200" "--steps 20 --warmup 5 --batch 4 --straggler-ticks 200" "--steps 20 --warmup 5 --straggler-ticks 200"; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2z3.json 2> gpurun_out/r2z3.err
    python - "args=[$a]" gpurun_out/r2z3.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
done 2>&1 | tee gpurun_out/r2z3_ttl.log
