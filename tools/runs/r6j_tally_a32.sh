#!/bin/bash
# the tally's answer bits through a 32-bit window per pass (one 64-bit shift per follower and pass, not per row) against the build
# before (variants/libsummerset_hip_mphead.so): the driver's command and the steady state, interleaved; then the MultiPaxos device tests
for rep in 1 2; do
  for v in mphead new; do
    if [ $v = new ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$v.so; fi
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r6j_${v}_${rep}.json 2>> gpurun_out/r6j.err
    timeout 200 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r6j_${v}_${rep}_steady.json 2>> gpurun_out/r6j.err
    python - <<P
import json
for f in ("", "_steady"):
    d = json.loads(open("gpurun_out/r6j_${v}_${rep}%s.json" % f).read().strip().splitlines()[-1])
    k = d.get("kernels") or {}
    print("$v $rep%s ms/tick %.4f" % (f, d["ms_per_step"]), "tally us %.2f" % d["roofline"]["avg_launch_us"], {n: round(x["avg_us"], 1) for n, x in k.items()})
P
  done
done
unset SUMMERSET_HIP_LIB
timeout 900 python -m pytest tests/test_mp_gpu.py tests/test_baseline_configs_gpu.py tests/test_zz_mp_wide_gpu.py -m gpu -q -x -p no:cacheprovider -k "not ep and not rsp and not config3 and not config4 and not config5" 2>&1 | tail -3
