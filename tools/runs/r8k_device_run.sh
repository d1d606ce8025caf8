#!/bin/bash
# round 5: the cooperative side kernel with 1 (shipped) / 2 / 3 listed groups per block from the first one on, and the per-lane side
# kernel of round 4 -- the driver's command, same call, two rounds
mkdir -p gpurun_out
V=$PWD/summerset_amd/variants
for i in 1 2; do
  for tag in shipped pack2 pack3 nocoop; do
    if [ $tag = shipped ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$V/libsummerset_hip_$tag.so; fi
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8k_${tag}_$i.json 2>> gpurun_out/r8k.err
  done
done
unset SUMMERSET_HIP_LIB
python - <<P
import json
for i in (1, 2):
    for n in ("shipped", "pack2", "pack3", "nocoop"):
        d = json.loads(open("gpurun_out/r8k_%s_%d.json" % (n, i)).read().strip().splitlines()[-1])
        print(n, i, "ms/tick %.4f  tally us %.1f frac %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
P
