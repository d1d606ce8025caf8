#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_raft_gpu.py tests/test_zz_craft_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -p no:cacheprovider -k "raft or craft" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --no-cpu --no-rs > gpurun_out/r2q_bench.json 2> gpurun_out/r2q_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2q_bench.json").read().strip().splitlines()[-1])
print("headline %.3e %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
print("raft", d["raft_quorum"]["roofline"]["frac"], d["raft_quorum"]["roofline"]["avg_launch_us"], d["raft_quorum"]["us_per_tick"], "%.3e" % d["raft_quorum"]["value"])
PY
done
