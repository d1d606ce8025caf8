#!/bin/bash
# EPaxos one-launch tick: instance records through non-temporal loads / stores (-DEP_NT variant) against the shipped build, interleaved
for rep in 1 2; do
  for v in base nt; do
    if [ $v = nt ]; then export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_epnt.so; else unset SUMMERSET_HIP_LIB; fi
    timeout 400 python bench.py --leg epaxos_cluster > gpurun_out/r6a_${v}_${rep}.json 2>> gpurun_out/r6a.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6a_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep", {k: round(x["ms_per_tick"], 4) for k, x in d.items() if isinstance(x, dict) and "ms_per_tick" in x})
P
  done
done
