#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_zz_ep_cluster_gpu.py tests/test_zz_hb_gpu.py tests/test_zz_skv_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
( time timeout 900 python bench.py > gpurun_out/r2n_bench.json 2> gpurun_out/r2n_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2n_bench.json").read().strip().splitlines()[-1])
print("headline %.3e %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
print("raft", d["raft_quorum"]["roofline"]["frac"], d["raft_quorum"]["roofline"]["avg_launch_us"], d["raft_quorum"]["us_per_tick"], "%.3e" % d["raft_quorum"]["value"])
print("ep", d["epaxos_fast_quorum"]["roofline"]["frac"], d["epaxos_fast_quorum"]["roofline"]["avg_launch_us"], d["epaxos_fast_quorum"].get("propose_kernel_us"))
print("ep cluster", d.get("epaxos_cluster"))
print("rs", d["rs_encode"]["value"], d["rs_encode"]["roofline"]["frac"], d["rs_encode"]["one_launch_65536_codewords"])
print("rspaxos", d.get("rspaxos"))
PY
