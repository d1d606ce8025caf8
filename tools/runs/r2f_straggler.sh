#!/bin/bash
# the straggler side stream again, now that its fused tick kernel has no agent-scope fences (r2e): default workload
mkdir -p gpurun_out
for s in 0 1 2 4 8; do
  for i in 1 2; do
    timeout 200 python bench.py --no-cpu --no-rs --no-extra --fused 0 --straggler-ticks $s > gpurun_out/r2f_s${s}.json 2> gpurun_out/r2f_s${s}.err
    python - "straggler_ticks=$s" gpurun_out/r2f_s${s}.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-500:])
PY
  done
done 2>&1 | tee gpurun_out/r2f_straggler.log
timeout 600 python -m pytest tests/test_raft_gpu.py tests/test_zz_craft_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r2f_raft_tests.log
timeout 200 python bench.py --no-cpu --no-rs > gpurun_out/r2f_bench_extra.json 2> gpurun_out/r2f_bench_extra.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2f_bench_extra.json").read().strip().splitlines()[-1])
print("raft", d.get("raft_quorum", {}).get("roofline"), d.get("raft_quorum", {}).get("us_per_tick"))
PY
