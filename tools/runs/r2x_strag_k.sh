#!/bin/bash
# r2x: listed groups per straggler block (lanes 0..K-1 of every replica's wavefront, one block per CU), list capacity
mkdir -p gpurun_out
V=$PWD/summerset_amd/variants
for lib in "" $V/libsummerset_hip_k2.so $V/libsummerset_hip_k4.so $V/libsummerset_hip_k4c2048.so $V/libsummerset_hip_k8c2048.so $V/libsummerset_hip_k16c4096.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  for a in "" "--straggler-ticks 4" "--steps 20 --warmup 5" "--steps 20 --warmup 5"; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2x.json 2> gpurun_out/r2x.err
    python - "lib=$(basename "$lib") args=[$a]" gpurun_out/r2x.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
  done
done 2>&1 | tee gpurun_out/r2x_strag_k.log
