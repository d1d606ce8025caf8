#!/bin/bash
# the bulk round kernels' wavefronts at a raised issue priority (s_setprio 1 / 3, -DSMR_BULK_PRIO) beside the side launch: driver's command, same call
mkdir -p gpurun_out; R=$PWD
for i in 1 2; do
  for v in shipped prio1 prio3; do
    L=$R/summerset_amd/libsummerset_hip.so; [ $v != shipped ] && L=$R/summerset_amd/variants/libsummerset_hip_$v.so
    SUMMERSET_HIP_LIB=$L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r9h_${v}_$i.json 2>> gpurun_out/r9h.err
  done
done
python - <<P
import json
for i in (1, 2):
    for n in ("shipped", "prio1", "prio3"):
        try:
            d = json.loads(open("gpurun_out/r9h_%s_%d.json" % (n, i)).read().strip().splitlines()[-1])
            print(n, i, "ms/tick %.4f  tally us %.1f" % (d["ms_per_step"], d["roofline"]["avg_launch_us"]))
        except Exception as e:
            print(n, i, "unreadable", e)
P
tail -3 gpurun_out/r9h.err
