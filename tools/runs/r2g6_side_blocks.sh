#!/bin/bash
# r2g6: blocks x groups per block of the batch straggler kernel, default flags and the driver's, three runs each
mkdir -p gpurun_out
V=$PWD/summerset_amd/variants
for lib in "" $V/libsummerset_hip_b192k6.so $V/libsummerset_hip_b160k8.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  for a in "" "--steps 20 --warmup 5"; do
   for i in 1 2 3; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2g6.json 2> gpurun_out/r2g6.err
    python - "lib=$(basename "$lib") args=[$a]" gpurun_out/r2g6.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
   done
  done
done 2>&1 | tee gpurun_out/r2g6_side_blocks.log
