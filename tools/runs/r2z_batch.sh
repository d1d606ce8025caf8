#!/bin/bash
# r2z: smr_mp_run_ticks with the straggler list on (bench.py --batch): device tests of the mode, then A/B against the per-tick call
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mp_gpu.py -q -m gpu -p no:cacheprovider -k "batch or fused or bench_shape" 2>&1 | tail -3 | tee gpurun_out/r2z_tests.log
for a in "" "--batch 16" "--batch 8" "--batch 4" "--batch 16 --straggler-ticks 4" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --batch 16" "--steps 20 --warmup 5 --batch 16" "--batch 16 --timeouts 0" "--timeouts 0"; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2z.json 2> gpurun_out/r2z.err
    python - "args=[$a]" gpurun_out/r2z.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
done 2>&1 | tee gpurun_out/r2z_batch.log
