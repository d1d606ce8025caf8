#!/bin/bash
# sharded counters: the whole -m gpu suite, then the bench line (default and steady) with every leg
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r2r_gputests.log
timeout 600 python bench.py --no-cpu > gpurun_out/r2r_bench.json 2> gpurun_out/r2r_bench.err
timeout 300 python bench.py --no-cpu --no-rs --no-extra --timeouts 0 > gpurun_out/r2r_bench_steady.json 2>> gpurun_out/r2r_bench.err
timeout 300 python bench.py --no-cpu --no-rs --no-extra --steps 20 --warmup 5 > gpurun_out/r2r_bench_driver.json 2>> gpurun_out/r2r_bench.err
python - <<'PY'
import json
def last(f): return json.loads(open(f).read().strip().splitlines()[-1])
d = last("gpurun_out/r2r_bench.json")
for tag, x in (("default", d), ("steady", last("gpurun_out/r2r_bench_steady.json")), ("driver flags", last("gpurun_out/r2r_bench_driver.json"))):
    print(tag, "%.3e %.4f frac %.3f" % (x["value"], x["ms_per_step"], x["roofline"]["frac"]), {n: round(v.get("avg_us") or 0, 1) for n, v in x["kernels"].items()})
print("raft", d["raft_quorum"]["roofline"]["frac"], d["raft_quorum"]["roofline"]["avg_launch_us"], d["raft_quorum"]["us_per_tick"], "%.3e" % d["raft_quorum"]["value"])
print("ep", d["epaxos_fast_quorum"]["roofline"]["frac"], d["epaxos_fast_quorum"]["roofline"]["avg_launch_us"], d["epaxos_fast_quorum"].get("propose_kernel_us"))
print("ep cluster", d["epaxos_cluster"]["value"], d["epaxos_cluster"]["ms_per_tick"])
print("rspaxos", d["rspaxos"]["value"], d["rspaxos"]["ms_per_tick"])
PY
