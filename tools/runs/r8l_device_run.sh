#!/bin/bash
# round 5: batch length x straggler ttl re-swept on the cooperative side kernel (driver's command otherwise)
mkdir -p gpurun_out
for b in 8 12 16; do for t in 2 4 6; do
  timeout 300 python bench.py --gpus 1 --steps 48 --warmup 8 --repeats 5 --batch $b --straggler-ticks $t --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8l_b${b}_t${t}.json 2>> gpurun_out/r8l.err
done; done
python - <<P
import json
for b in (8, 12, 16):
    for t in (2, 4, 6):
        d = json.loads(open("gpurun_out/r8l_b%d_t%d.json" % (b, t)).read().strip().splitlines()[-1])
        print("batch %2d ttl %d  ms/tick %.4f  tally us %.1f" % (b, t, d["ms_per_step"], d["roofline"]["avg_launch_us"]))
P
