#!/bin/bash
# round 5: the R3 rest in the next tick's R1 launch ALWAYS (round 4 measured it slower beside the old side kernel, profiles/r5p) against quiet stretches only
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8s_quiet_only_$i.json 2>> gpurun_out/r8s.err
  SMR_MP_ALWAYS_DEFER_REST=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8s_always_$i.json 2>> gpurun_out/r8s.err
done
python - <<P
import json
for i in (1, 2, 3):
    for n in ("quiet_only", "always"):
        d = json.loads(open("gpurun_out/r8s_%s_%d.json" % (n, i)).read().strip().splitlines()[-1])
        print(n, i, "ms/tick %.4f  tally us %.1f" % (d["ms_per_step"], d["roofline"]["avg_launch_us"]))
P
