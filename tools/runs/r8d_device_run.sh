#!/bin/bash
# round 5: the EPaxos cluster's shared per-key table (one 128-byte line per (group, key) for the five replicas) -- its tests, then the
# leg with the shared table and, same call, with private tables (SMR_EP_PRIVATE_HC=1)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zz_ep_cluster_gpu.py tests/test_zzz_ep_cluster_fused_gpu.py tests/test_zz_ep_exec_gpu.py tests/test_zzy_spread_ep_gpu.py tests/test_zzz_example_ep_gpu.py "tests/test_baseline_configs_gpu.py" -k "ep or config4 or config5" -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r8d_tests.log
tail -3 gpurun_out/r8d_tests.log
for i in 1 2; do
  timeout 300 python bench.py --leg epaxos_cluster > gpurun_out/r8d_leg_shared_$i.json 2>> gpurun_out/r8d.err
  SMR_EP_PRIVATE_HC=1 timeout 300 python bench.py --leg epaxos_cluster > gpurun_out/r8d_leg_private_$i.json 2>> gpurun_out/r8d.err
done
python - <<P
import json
for n in ("shared_1", "private_1", "shared_2", "private_2"):
    d = json.loads(open("gpurun_out/r8d_leg_%s.json" % n).read().strip().splitlines()[-1])
    a, b = d["one_call_per_tick"], d["one_call_per_tick_phase_by_phase"]
    print(n, "loops' order ms %.4f (device median %.1f us)" % (a["ms_per_tick"], a["tick_us_device_median"]), "| phase by phase ms %.4f (device median %.1f, min %.1f us)" % (b["ms_per_tick"], b["tick_us_device_median"], b["tick_us_device_min"]))
P
