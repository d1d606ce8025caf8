#!/bin/bash
# the one-launch EPaxos tick with its messages through LDS: the leg (both orders, execution on / off), its device tests
timeout 400 python bench.py --leg epaxos_cluster > gpurun_out/${1:-r5g}_leg_epaxos_cluster.json 2> gpurun_out/${1:-r5g}.err
python - <<P
import json
d = json.loads(open("gpurun_out/${1:-r5g}_leg_epaxos_cluster.json").read().strip().splitlines()[-1])
def walk(x, pre=""):
    if isinstance(x, dict):
        for k, v in x.items():
            if isinstance(v, (dict,)): walk(v, pre + k + ".")
            elif isinstance(v, (int, float)) and ("ms" in k or "us" in k or k in ("value", "frac")): print(pre + k, round(v, 4))
walk(d)
P
timeout 900 python -m pytest tests/test_zzz_ep_cluster_fused_gpu.py tests/test_zz_ep_cluster_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q -x -p no:cacheprovider -k "ep or config4 or config5" 2>&1 | tail -3
