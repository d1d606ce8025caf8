#!/bin/bash
# r2z2: listed groups per block in the batch straggler kernel (STRAG_BATCH_K) x ticks per batch
mkdir -p gpurun_out
V=$PWD/summerset_amd/variants
timeout 600 python -m pytest tests/test_mp_gpu.py -q -m gpu -p no:cacheprovider -k "batch" 2>&1 | tail -2
for lib in "" $V/libsummerset_hip_bk1.so $V/libsummerset_hip_bk2.so $V/libsummerset_hip_bk8.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  for a in "--batch 16" "--batch 8" "--steps 20 --warmup 5 --batch 16" "--steps 20 --warmup 5 --batch 8" "--steps 20 --warmup 5 --batch 4"; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2z2.json 2> gpurun_out/r2z2.err
    python - "lib=$(basename "$lib") args=[$a]" gpurun_out/r2z2.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
  done
done 2>&1 | tee gpurun_out/r2z2_batch_k.log
