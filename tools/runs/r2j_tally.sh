#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mp_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -p no:cacheprovider -k "not fused" 2>&1 | tail -2
for lib in "" $PWD/summerset_amd/variants/libsummerset_hip_tallyc4.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  for a in "" "--timeouts 0" "--steps 20 --warmup 5"; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2j.json 2> gpurun_out/r2j.err
    python - "lib=$(basename "$lib") args=[$a]" gpurun_out/r2j.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-500:])
PY
  done
done 2>&1 | tee gpurun_out/r2j_tally.log
