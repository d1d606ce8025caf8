#!/bin/bash
# r3p: the EPaxos engine on instance RECORDS (16-byte words instead of a plane per field): parity tests, the cluster leg (execution on / off), loads per wavefront
TAG=${1:-r3p}
mkdir -p gpurun_out
V=$PWD/summerset_amd/variants
run() { timeout 200 python bench.py --leg epaxos_cluster 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('one_call_per_tick','one_call_per_tick_per_handler_launches'):
    o=d.get(k,{}); print('   ',k, 'ms/tick', round(o.get('ms_per_tick',0),4), 'device median us', round(o.get('tick_us_device_median',0),1), 'min', round(o.get('tick_us_device_min',0),1), o.get('error',''))
"; }
{
timeout 600 python -m pytest tests/test_zz_ep_cluster_gpu.py tests/test_zzz_ep_cluster_fused_gpu.py tests/test_zzz_example_ep_gpu.py tests/test_ep_gpu.py tests/test_zz_ep_exec_gpu.py tests/test_zz_ep_recovery_gpu.py tests/test_zzz_ep_recovery_exec_gpu.py tests/test_zzy_spread_ep_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q -p no:cacheprovider -k "ep or epaxos" 2>&1 | tail -3
for lib in "" $V/libsummerset_hip_epc_w4.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  echo "lib=$(basename "$lib") execute=1"; run
  echo "lib=$(basename "$lib") execute=0"; SMR_EPC_EXECUTE=0 run
done
unset SUMMERSET_HIP_LIB
bash tools/runs/r3o_epc_loads.sh > /dev/null 2>&1; cat gpurun_out/r3o_epc_loads.txt | cut -c1-330
} 2>&1 | tee gpurun_out/${TAG}_epc_records.log
