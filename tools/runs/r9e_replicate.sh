#!/bin/bash
# smr_raft_cluster_replicate: its device tests, the craft_payload leg with it (default) -- and the payload stores' tests again
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_raft_gpu.py tests/test_zz_craft_payload_gpu.py tests/test_zz_craft_follower_gpu.py tests/test_zz_craft_gpu.py \
  tests/test_baseline_configs_gpu.py -m gpu -q -k "raft or craft or Raft" -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r9e_tests.log
tail -3 gpurun_out/r9e_tests.log
for k in 1 2; do
  timeout 300 python bench.py --leg craft_payload > gpurun_out/r9e_leg_craft_payload_$k.json 2>> gpurun_out/r9e.err
done
python - <<P
import json
for k in (1, 2):
    try:
        d = json.loads(open("gpurun_out/r9e_leg_craft_payload_%d.json" % k).read().strip().splitlines()[-1])
        d = d.get("craft_payload", d)
        print("craft_payload", k, "ms/tick %.4f" % d["ms_per_tick"], "engines %.4f" % d.get("engine_only_ms_per_tick", 0), "bytes path %.4f" % d.get("bytes_path_ms_per_tick", 0), "verified", d.get("verified"))
    except Exception as e:
        print(k, "unreadable", e)
P
tail -3 gpurun_out/r9e.err
