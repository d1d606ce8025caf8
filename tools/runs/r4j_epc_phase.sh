#!/bin/bash
# r4j: the one-launch EPaxos tick with the leaders' steps phase by phase: its parity tests on the device, the epaxos_cluster leg (all modes)
mkdir -p gpurun_out
R=$PWD; export PYTHONPATH=$R
{ timeout 900 python -m pytest tests/test_zzz_ep_cluster_fused_gpu.py tests/test_zz_ep_cluster_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout 600 python bench.py --leg epaxos_cluster 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r4j_leg_epaxos_cluster.json
python - <<'P'
import json
d = json.load(open("gpurun_out/r4j_leg_epaxos_cluster.json"))
for k, v in d.items():
    if isinstance(v, dict):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, list))})
    else:
        print(k, v if not isinstance(v, float) else round(v, 4))
P
} 2>&1 | tee gpurun_out/r4j.log
