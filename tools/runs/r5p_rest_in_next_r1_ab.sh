#!/bin/bash
# same-call A/B: the previous build (variants/libsummerset_hip_prev.so) against the shipped one, the driver's command and the
# steady state, twice each (box noise), then the MultiPaxos device tests on the shipped build
R=$PWD
B="--no-l2 --no-extra --no-rs --no-cpu"
for i in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then export SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_prev.so; else unset SUMMERSET_HIP_LIB; fi
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r5p_${v}_driver_$i.json 2>> gpurun_out/r5p.err
    timeout 200 python bench.py --timeouts 0 $B > gpurun_out/r5p_${v}_steady_$i.json 2>> gpurun_out/r5p.err
  done
done
unset SUMMERSET_HIP_LIB
python - <<P
import json
for v in ("prev", "new"):
    for k in ("driver", "steady"):
        for i in (1, 2):
            try:
                d = json.loads(open("gpurun_out/r5p_%s_%s_%d.json" % (v, k, i)).read().strip().splitlines()[-1])
                ks = d["kernels"]
                print(v, k, i, "ms/tick %.4f" % d["ms_per_step"], " ".join("%s %.1f" % (n.split("_")[0], ks[n]["avg_us"]) for n in ks), "tally frac %.3f" % d["roofline"]["frac"])
            except Exception as e:
                print(v, k, i, "unreadable", e)
P
timeout 900 python -m pytest tests/test_mp_gpu.py tests/test_baseline_configs_gpu.py tests/test_zz_mp_wide_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
