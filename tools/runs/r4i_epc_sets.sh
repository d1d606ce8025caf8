#!/bin/bash
# r4i: the one-launch EPaxos tick with 2 / 3 sets of 64 groups per block (10 / 15 wavefronts) at 3 / 4 wavefronts per SIMD, against the shipped build
mkdir -p gpurun_out
R=$PWD; export PYTHONPATH=$R
for v in "" epc_s2w3 epc_s3w4; do
  [ -n "$v" ] && export SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_$v.so
  echo "== ${v:-shipped}"
  [ -n "$v" ] && timeout 600 python -m pytest tests/test_zz_ep_cluster_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
  timeout 300 python bench.py --leg epaxos_cluster 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
o=d.get('one_call_per_tick',d)
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in o.items() if not isinstance(v,(dict,list))})"
done 2>&1 | tee gpurun_out/r4i.log
