#!/bin/bash
# r2g5: the driver's flags (--steps 20 --warmup 5, 26 leader changes per tick), three runs each: blocks x groups per block of the batch
# straggler kernel, batch size, and one call per tick
mkdir -p gpurun_out
V=$PWD/summerset_amd/variants
for cfg in "|--batch 8" "|--batch 5" "|--batch 0 --straggler-ticks 8" "$V/libsummerset_hip_b128k8.so|--batch 8" "$V/libsummerset_hip_b192k6.so|--batch 8" "$V/libsummerset_hip_b128k8.so|--batch 5"; do
  lib=${cfg%%|*}; a=${cfg#*|}
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  for i in 1 2 3; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra --steps 20 --warmup 5 $a > gpurun_out/r2g5.json 2> gpurun_out/r2g5.err
    python - "lib=$(basename "$lib") args=[$a]" gpurun_out/r2g5.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
  done
done 2>&1 | tee gpurun_out/r2g5_driver_flags.log
