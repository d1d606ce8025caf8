#!/bin/bash
# r4q: the one-launch EPaxos tick, both orders, with and without dependency-graph execution
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for ex in 1 0; do
  SMR_EPC_EXECUTE=$ex timeout 600 python bench.py --leg epaxos_cluster 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('one_call_per_tick','one_call_per_tick_phase_by_phase'):
    o=d[k]; print('execute=$ex', k, 'device median us', round(o['tick_us_device_median'],1), 'min', round(o['tick_us_device_min'],1), 'same commits', o['same_commits_as_the_driver_loop'])"
done 2>&1 | tee gpurun_out/r4q.log
