#!/bin/bash
# the side kernel's geometry (listed groups per block K x blocks) re-swept on round 4's kernels: the driver's command, interleaved
for rep in 1 2; do
  for v in base k8b192 k12b192 k6b256 k10b128 k16b96; do
    if [ $v = base ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$v.so; fi
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r6e_${v}_${rep}.json 2>> gpurun_out/r6e.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6e_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep ms/tick %.4f" % d["ms_per_step"], "regions", [round(x, 4) for x in d["timed_regions"]["ms_per_step"]])
P
  done
done
