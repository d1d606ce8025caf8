#!/bin/bash
# mp_round_deliver_all's register budget: __launch_bounds__(256, MP_R2_MINW) 5 (shipped: 96 VGPRs, 852 B of scratch) against 4 / 3 / 2, driver's command + steady state, same call
mkdir -p gpurun_out; R=$PWD
for i in 1 2; do
  for v in shipped r2w4 r2w3 r2w2; do
    L=$R/summerset_amd/libsummerset_hip.so; [ $v != shipped ] && L=$R/summerset_amd/variants/libsummerset_hip_$v.so
    SUMMERSET_HIP_LIB=$L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r9i_${v}_$i.json 2>> gpurun_out/r9i.err
    SUMMERSET_HIP_LIB=$L timeout 300 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r9i_${v}_steady_$i.json 2>> gpurun_out/r9i.err
  done
done
python - <<P
import json
for i in (1, 2):
    for n in ("shipped", "r2w4", "r2w3", "r2w2"):
        try:
            d = json.loads(open("gpurun_out/r9i_%s_%d.json" % (n, i)).read().strip().splitlines()[-1])
            s = json.loads(open("gpurun_out/r9i_%s_steady_%d.json" % (n, i)).read().strip().splitlines()[-1])
            print(n, i, "ms/tick %.4f  steady %.4f" % (d["ms_per_step"], s["ms_per_step"]))
        except Exception as e:
            print(n, i, "unreadable", e)
P
tail -3 gpurun_out/r9i.err
