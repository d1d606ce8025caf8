#!/bin/bash
# r2w: the straggler kernel with a group's replicas on neighbouring lanes (one wavefront per listed group, STRAG_GPW groups per
# wavefront), list capacities 1024..8192; plus the lease manager's device tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zz_lease_gpu.py tests/test_mp_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r2w_tests.log
V=$PWD/summerset_amd/variants
for lib in "" $V/libsummerset_hip_c2048.so $V/libsummerset_hip_c4096.so $V/libsummerset_hip_g2c4096.so $V/libsummerset_hip_g4c4096.so $V/libsummerset_hip_g4c8192.so $V/libsummerset_hip_g8c8192.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  for a in "" "--straggler-ticks 4" "--straggler-ticks 16" "--steps 20 --warmup 5"; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2w.json 2> gpurun_out/r2w.err
    python - "lib=$(basename "$lib") args=[$a]" gpurun_out/r2w.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
  done
done 2>&1 | tee gpurun_out/r2w_strag.log
