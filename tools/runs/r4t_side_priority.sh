#!/bin/bash
# r4t: the straggler side stream at the device's lowest stream priority (variant build) against the shipped build, driver flags, twice each
mkdir -p gpurun_out
R=$PWD
for rep in 1 2; do for v in "" side_lowprio; do
  if [ -n "$v" ]; then export SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_$v.so; else unset SUMMERSET_HIP_LIB; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('${v:-shipped}', 'ms/step %.4f' % d['ms_per_step'], ' '.join('%s %.1f' % (a, b['avg_us']) for a, b in k.items()))"
done; done 2>&1 | tee gpurun_out/r4t.log
