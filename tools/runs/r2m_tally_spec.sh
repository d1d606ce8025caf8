#!/bin/bash
# r2m: the tally's speculative leader loads (TALLY_SPEC) and per-wavefront flag bytes (TALLY_WAVEFLAGS), same-call A/B:
# default workload and steady state, shipped build first and last
mkdir -p gpurun_out
V=$PWD/summerset_amd/variants
for lib in "" $V/libsummerset_hip_spec.so $V/libsummerset_hip_wf.so $V/libsummerset_hip_specwf.so ""; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  for a in "" "--timeouts 0"; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2m.json 2> gpurun_out/r2m.err
    python - "lib=$(basename "$lib") args=[$a]" gpurun_out/r2m.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-500:])
PY
  done
done 2>&1 | tee gpurun_out/r2m_tally_spec.log
