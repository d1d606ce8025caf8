#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_spread_mp.py tests/test_raft_gpu.py tests/test_zz_craft_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -p no:cacheprovider -k "spread or raft or craft" 2>&1 | tail -2
for n in 4 8; do timeout 300 python bench.py --layout spread --spread-ranks $n --steps 24 --warmup 6 > gpurun_out/r2o_spread$n.json 2> gpurun_out/r2o_spread.err; python - gpurun_out/r2o_spread$n.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("spread", d["config"]["spread_ranks"], "%.3e" % d["value"], d["ms_per_step"], d["exchange"]["bytes_sent_per_tick_per_rank"])
PY
done
timeout 300 python bench.py --no-cpu --no-rs > gpurun_out/r2o_bench.json 2> gpurun_out/r2o_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2o_bench.json").read().strip().splitlines()[-1])
print("headline %.3e %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
print("raft", d["raft_quorum"]["roofline"]["frac"], d["raft_quorum"]["roofline"]["avg_launch_us"], d["raft_quorum"]["us_per_tick"], "%.3e" % d["raft_quorum"]["value"])
PY
