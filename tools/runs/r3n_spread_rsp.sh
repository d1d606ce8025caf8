#!/bin/bash
# r3n: RSPaxos L2 device-resident on the device (virtual ranks), the ground-truth ingest test, the Raft follower after the ring-guard change
mkdir -p gpurun_out
{ timeout 600 python -m pytest tests/test_spread_rsp.py tests/test_zz_wire_ingest_gpu.py tests/test_raft_gpu.py tests/test_zz_craft_follower_gpu.py tests/test_rs_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
for n in 4 8; do
timeout 300 python bench.py --layout spread-rspaxos --spread-ranks $n --steps 24 --warmup 6 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r3n_bench_spread_rspaxos$n.json
python - <<P
import json
d=json.loads(open("gpurun_out/r3n_bench_spread_rspaxos$n.json").read())
print("ranks $n: ms/tick", round(d["ms_per_step"],4), "slots/s", d["value"], d["exchange"])
P
done
} 2>&1 | tee gpurun_out/r3n.log
