#!/bin/bash
# same-call A/B of the tally's rows per pass: tc4 (4 rows x 2 passes, round 2's shape), the shipped build (8 rows, one pass, 102 VGPRs),
# tc8w5 (8 rows, capped at 96 VGPRs: 28 B of scratch); driver command + steady, twice each
R=$PWD
B="--no-l2 --no-extra --no-rs --no-cpu"
for i in 1 2; do
  for v in tc4 new tc8w5; do
    if [ $v = new ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_$v.so; fi
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r5e_${v}_driver_$i.json 2>> gpurun_out/r5e.err
    timeout 200 python bench.py --timeouts 0 $B > gpurun_out/r5e_${v}_steady_$i.json 2>> gpurun_out/r5e.err
  done
done
unset SUMMERSET_HIP_LIB
python - <<P
import json
for v in ("tc4", "new", "tc8w5"):
    for k in ("driver", "steady"):
        for i in (1, 2):
            try:
                d = json.loads(open("gpurun_out/r5e_%s_%s_%d.json" % (v, k, i)).read().strip().splitlines()[-1])
                ks = d["kernels"]
                print(v, k, i, "ms/tick %.4f" % d["ms_per_step"], " ".join("%s %.1f" % (n.split("_")[0], ks[n]["avg_us"]) for n in ks), "tally frac %.3f" % d["roofline"]["frac"])
            except Exception as e:
                print(v, k, i, "unreadable", e)
P
timeout 900 python -m pytest tests/test_mp_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q -x -p no:cacheprovider -k "mp or multipaxos or headline or config1" 2>&1 | tail -3
