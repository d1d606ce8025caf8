#!/bin/bash
# the driver's command (20 steps per region) at batch lengths that divide 20 against the default 8 (8 + 8 + 4 per region), interleaved
for rep in 1 2; do
  for b in 8 10 5 7; do
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --batch $b --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r6k_b${b}_${rep}.json 2>> gpurun_out/r6k.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6k_b${b}_${rep}.json").read().strip().splitlines()[-1])
print("batch $b rep $rep ms/tick %.4f" % d["ms_per_step"], [round(x, 4) for x in d["timed_regions"]["ms_per_step"]])
P
  done
done
