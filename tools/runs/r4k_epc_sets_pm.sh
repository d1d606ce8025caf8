#!/bin/bash
# r4k: phase by phase with two sets of 64 groups per block (10 wavefronts over 4 SIMDs instead of 5) at 168 VGPRs
R=$PWD; export PYTHONPATH=$R
export SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_epc_s2w3.so
mkdir -p gpurun_out
{ timeout 600 python -m pytest tests/test_zzz_ep_cluster_fused_gpu.py -m gpu -q -x -p no:cacheprovider -k phase 2>&1 | tail -2
timeout 600 python bench.py --leg epaxos_cluster 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('one_call_per_tick','one_call_per_tick_phase_by_phase'):
    o=d[k]; print(k,{a:(round(b,4) if isinstance(b,float) else b) for a,b in o.items() if not isinstance(b,(dict,list,str))})"
} 2>&1 | tee gpurun_out/r4k.log
