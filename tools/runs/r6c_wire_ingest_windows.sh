#!/bin/bash
# the one-launch smr_wire_ingest_mp at window sizes 128 / 256 / 608 bytes against the two-launch build (128-byte windows)
for rep in 1 2; do
  for v in wihead wi128 wi256 new; do
    if [ $v = new ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$v.so; fi
    timeout 300 python bench.py --leg wire_ingest > gpurun_out/r6c_${v}_${rep}.json 2>> gpurun_out/r6c.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6c_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep call_us", round(d["call_us"], 1), "frac", round(d["roofline"]["frac"], 3))
P
  done
done
