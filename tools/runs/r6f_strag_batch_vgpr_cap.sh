#!/bin/bash
# the side kernel under a VGPR cap (3 / 4 wavefronts per SIMD = 168 / 128 registers, spills go to scratch): the driver's command, interleaved
for rep in 1 2; do
  for v in base sw3 sw4; do
    if [ $v = base ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$v.so; fi
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r6f_${v}_${rep}.json 2>> gpurun_out/r6f.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6f_${v}_${rep}.json").read().strip().splitlines()[-1])
k = d.get("kernels") or {}
print("$v $rep ms/tick %.4f" % d["ms_per_step"], {n: round(x["avg_us"], 1) for n, x in k.items()})
P
  done
done
