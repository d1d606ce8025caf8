#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_mp_gpu.py -q -m gpu -p no:cacheprovider -k "leader_change or bench_shape" 2>&1 | tail -2
for a in "" "--steps 20 --warmup 5" "--straggler-ticks 6" "--straggler-ticks 12"; do
  for i in 1 2; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2i.json 2> gpurun_out/r2i.err
    python - "args=[$a]" gpurun_out/r2i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-500:])
PY
  done
done 2>&1 | tee gpurun_out/r2i_quick.log
