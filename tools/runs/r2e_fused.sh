#!/bin/bash
# device run of the fused tick kernel: parity, then the bench line fused (16 / 4 / 1 ticks per launch) and per-round;
# the shipped build (FUSED_MINW=3: 168 VGPRs, spills) and the variant with 229 VGPRs and no spills
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mp_gpu.py -q -m gpu -p no:cacheprovider -k "fused or leader_change" 2>&1 | tail -5 | tee gpurun_out/r2e_tests.log
for lib in "" $PWD/summerset_amd/variants/libsummerset_hip_minw2.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  for f in 16 4 1 0; do
    for w in "--timeouts 0.01" "--timeouts 0"; do
        timeout 200 python bench.py --no-cpu --no-rs --no-extra --fused $f $w > gpurun_out/r2e_f${f}.json 2> gpurun_out/r2e_f${f}.err
        python - "lib=$(basename "$lib") fused=$f $w" gpurun_out/r2e_f${f}.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], d["roofline"].get("avg_launch_us"))
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-500:])
PY
    done
  done
done 2>&1 | tee gpurun_out/r2e_fused.log
