#!/bin/bash
# round 5: the CRaft payload leg (first device run), the RSPaxos payload leg after "plan stores what changed", their device tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zz_craft_payload_gpu.py tests/test_zz_rsp_payload_gpu.py tests/test_zzzz_rsp_emit_accepts_gpu.py tests/test_baseline_configs_gpu.py -k "craft or payload or emit" -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r8f_tests.log
tail -3 gpurun_out/r8f_tests.log
timeout 300 python bench.py --leg craft_payload > gpurun_out/r8f_leg_craft_payload.json 2> gpurun_out/r8f.err
timeout 300 python bench.py --leg rspaxos_payload > gpurun_out/r8f_leg_rspaxos_payload.json 2>> gpurun_out/r8f.err
python - <<P
import json
for n in ("craft_payload", "rspaxos_payload"):
    try:
        d = json.loads(open("gpurun_out/r8f_leg_%s.json" % n).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(n, "ms/tick %.4f engine %.4f bytes %.4f frac %.3f on 8d %.3f verified %s" % (d["ms_per_tick"], d["engine_only_ms_per_tick"], d["bytes_path_ms_per_tick"], r["frac"], r["frac_on_survey_8d_bytes"], d["verified"]), d["counters"])
    except Exception as e:
        print(n, "unreadable", e)
P
tail -3 gpurun_out/r8f.err
