#!/bin/bash
# votes as aliases of the reqs row's shards (csrc/rsp_payload.hip): the stores' device tests, then the two payload legs
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zz_rsp_payload_gpu.py tests/test_zzz_example_rsp_payload_gpu.py tests/test_zzzz_rsp_emit_accepts_gpu.py \
  tests/test_zz_craft_payload_gpu.py tests/test_zz_rsp_bytes_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q -k "payload or emit or bytes or craft" \
  -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r9b_tests.log
tail -3 gpurun_out/r9b_tests.log
for k in 1 2; do
  timeout 300 python bench.py --leg rspaxos_payload > gpurun_out/r9b_leg_rspaxos_payload_$k.json 2>> gpurun_out/r9b.err
  timeout 300 python bench.py --leg craft_payload > gpurun_out/r9b_leg_craft_payload_$k.json 2>> gpurun_out/r9b.err
done
python - <<P
import json
for leg in ("rspaxos_payload", "craft_payload"):
    for k in (1, 2):
        try:
            d = json.loads(open("gpurun_out/r9b_leg_%s_%d.json" % (leg, k)).read().strip().splitlines()[-1])
            d = d.get(leg, d)
            print(leg, k, "ms/tick %.4f" % d["ms_per_tick"], "bytes path %.4f" % d.get("bytes_path_ms_per_tick", 0), "frac %.3f" % d["roofline"]["frac"], "verified", d.get("verified"))
        except Exception as e:
            print(leg, k, "unreadable", e)
P
tail -5 gpurun_out/r9b.err
