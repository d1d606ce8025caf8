#!/bin/bash
# r2g1: register caps on the bulk round kernels (min wavefronts per SIMD via __launch_bounds__): more resident wavefronts against spills
mkdir -p gpurun_out
V=$PWD/summerset_amd/variants
for lib in "" $V/libsummerset_hip_r2w4.so $V/libsummerset_hip_r2w5.so $V/libsummerset_hip_r1w5.so $V/libsummerset_hip_tw6.so $V/libsummerset_hip_allw.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  for a in "" "--timeouts 0" "--steps 20 --warmup 5"; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2g1.json 2> gpurun_out/r2g1.err
    python - "lib=$(basename "$lib") args=[$a]" gpurun_out/r2g1.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
  done
done 2>&1 | tee gpurun_out/r2g1_occupancy.log
