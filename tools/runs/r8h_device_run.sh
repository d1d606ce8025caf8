#!/bin/bash
# round 5: the EPaxos tick with ONE set of 64 groups per block (EPC_SETS=1: 5 wavefronts, 76 KB of LDS, two blocks per CU) against the
# shipped two sets per block -- same call, alternating
mkdir -p gpurun_out
V=$PWD/summerset_amd/variants/libsummerset_hip_epsets1.so
for i in 1 2; do
  timeout 300 python bench.py --leg epaxos_cluster > gpurun_out/r8h_leg_sets2_$i.json 2>> gpurun_out/r8h.err
  SUMMERSET_HIP_LIB=$V timeout 300 python bench.py --leg epaxos_cluster > gpurun_out/r8h_leg_sets1_$i.json 2>> gpurun_out/r8h.err
done
python - <<P
import json
for n in ("sets2_1", "sets1_1", "sets2_2", "sets1_2"):
    d = json.loads(open("gpurun_out/r8h_leg_%s.json" % n).read().strip().splitlines()[-1])
    a, b = d["one_call_per_tick"], d["one_call_per_tick_phase_by_phase"]
    print(n, "loops' order device median %.1f us" % a["tick_us_device_median"], "| phase by phase device median %.1f, min %.1f us" % (b["tick_us_device_median"], b["tick_us_device_min"]))
P
