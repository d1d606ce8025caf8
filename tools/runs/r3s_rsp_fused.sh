#!/bin/bash
# r3s: smr_rsp_cluster_steady_tick (the RSPaxos steady tick in one launch): parity tests, the config-4 leg both ways
mkdir -p gpurun_out
{ timeout 600 python -m pytest tests/test_zz_rsp_steady_gpu.py tests/test_zz_rsp_gpu.py tests/test_spread_rsp.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for m in "" 1; do
  if [ -n "$m" ]; then export SMR_RSP_CALL_BY_CALL=1; else unset SMR_RSP_CALL_BY_CALL; fi
  timeout 300 python bench.py --leg rspaxos 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r3s_leg_rspaxos$m.json
  python - <<P
import json
d=json.loads(open("gpurun_out/r3s_leg_rspaxos$m.json").read())
print("call_by_call=$m", "eager ms/tick", round(d["eager"]["ms_per_tick"],4), "graph", d["graph"].get("ms_per_tick"), d["graph"].get("error"), "value", d["value"], "GiB/s", d["rs_payload_GiBps"], "roofline", d.get("roofline",{}).get("frac"))
P
done
} 2>&1 | tee gpurun_out/r3s.log
