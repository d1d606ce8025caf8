#!/bin/bash
# the slot model of DESIGN 9.1: a tally of <= 80 / <= 72 VGPRs (TALLY_MINW 6 / 7) fits 10 / 13 wavefronts on a CU that holds a side block, the shipped 87 fits 9
mkdir -p gpurun_out; R=$PWD
for i in 1 2; do
  for v in shipped tw6 tw7; do
    L=$R/summerset_amd/libsummerset_hip.so; [ $v != shipped ] && L=$R/summerset_amd/variants/libsummerset_hip_$v.so
    SUMMERSET_HIP_LIB=$L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r9k_${v}_$i.json 2>> gpurun_out/r9k.err
    SUMMERSET_HIP_LIB=$L timeout 300 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r9k_${v}_steady_$i.json 2>> gpurun_out/r9k.err
  done
done
python - <<P
import json
for i in (1, 2):
    for n in ("shipped", "tw6", "tw7"):
        try:
            d = json.loads(open("gpurun_out/r9k_%s_%d.json" % (n, i)).read().strip().splitlines()[-1])
            s = json.loads(open("gpurun_out/r9k_%s_steady_%d.json" % (n, i)).read().strip().splitlines()[-1])
            print(n, i, "ms/tick %.4f tally %.1f us | steady %.4f tally %.1f us" % (d["ms_per_step"], d["roofline"]["avg_launch_us"], s["ms_per_step"], s["roofline"]["avg_launch_us"]))
        except Exception as e:
            print(n, i, "unreadable", e)
P
tail -3 gpurun_out/r9k.err
