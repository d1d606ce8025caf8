#!/bin/bash
# r2o: the same tick over 4x / 8x the groups of the headline config (one launch = more block generations): where the
# round kernels go once a launch is no longer one generation of blocks
mkdir -p gpurun_out
for a in "--groups 65536 --timeouts 0" "--groups 262144 --timeouts 0" "--groups 524288 --timeouts 0" "--groups 262144"; do
  timeout 300 python bench.py --no-cpu --no-rs --no-extra $a > gpurun_out/r2o.json 2> gpurun_out/r2o.err
  python - "args=[$a]" gpurun_out/r2o.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], "whole_tick frac_alg %.3f" % d["roofline"]["whole_tick"]["frac_alg"], {n: round(v.get("avg_us") or 0, 1) for n, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "bench failed:", e, open(sys.argv[2].replace(".json", ".err")).read()[-500:])
PY
done 2>&1 | tee gpurun_out/r2o_groups.log
