#!/bin/bash
# EPaxos tick: replies of a decided instance not stored (new) and, on top, a barrier per set of 5 wavefronts instead of the block's
# (-DEPC_SET_BARRIER variant), against the build before (ephead); interleaved; then the device tests on both new builds
for rep in 1 2; do
  for v in ephead new epsetbar; do
    if [ $v = new ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$v.so; fi
    timeout 400 python bench.py --leg epaxos_cluster > gpurun_out/r6g_${v}_${rep}.json 2>> gpurun_out/r6g.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6g_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep", {k: round(x["ms_per_tick"], 4) for k, x in d.items() if isinstance(x, dict) and "ms_per_tick" in x})
P
  done
done
for v in new epsetbar; do
  if [ $v = new ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$v.so; fi
  echo "tests on $v"; timeout 600 python -m pytest tests/test_zzz_ep_cluster_fused_gpu.py tests/test_zz_ep_cluster_gpu.py tests/test_ep_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q -x -p no:cacheprovider -k "ep or config5" 2>&1 | tail -2
done
