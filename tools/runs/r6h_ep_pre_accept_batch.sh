#!/bin/bash
# EPaxos one-launch tick: an acceptor takes the four PreAccepts of a tick in two rounds of loads (ep_pre_accept_batch), against the build before (variants/libsummerset_hip_ephead.so),
# interleaved in one call; then the engine's device tests on the new build
for rep in 1 2; do
  for v in head new; do
    if [ $v = head ]; then export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_ephead.so; else unset SUMMERSET_HIP_LIB; fi
    timeout 400 python bench.py --leg epaxos_cluster > gpurun_out/r6h_${v}_${rep}.json 2>> gpurun_out/r6h.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6h_${v}_${rep}.json").read().strip().splitlines()[-1])
out = []
def walk(x, pre=""):
    if isinstance(x, dict):
        for k, v in x.items():
            if isinstance(v, dict): walk(v, pre + k + ".")
            elif isinstance(v, (int, float)) and ("ms_per" in k or "us_per" in k): out.append((pre + k, round(v, 4)))
walk(d)
print("$v $rep", out)
P
  done
done
unset SUMMERSET_HIP_LIB
timeout 900 python -m pytest tests/test_zzz_ep_cluster_fused_gpu.py tests/test_zz_ep_cluster_gpu.py tests/test_ep_gpu.py tests/test_zz_ep_exec_gpu.py tests/test_zz_ep_recovery_gpu.py tests/test_zzz_ep_recovery_exec_gpu.py tests/test_zzy_spread_ep_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q -x -p no:cacheprovider -k "ep or config5" 2>&1 | tail -3
