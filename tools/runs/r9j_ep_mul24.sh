#!/bin/bash
# EPaxos index arithmetic through 24-bit multiplies (-DEP_MUL24: 387 -> 37 quarter-rate v_mul_lo_u32 in the tick kernel): the leg, same call, and the device tests on the variant
mkdir -p gpurun_out; R=$PWD
V=$R/summerset_amd/variants/libsummerset_hip_mul24.so
for i in 1 2; do
  timeout 300 python bench.py --leg epaxos_cluster > gpurun_out/r9j_shipped_$i.json 2>> gpurun_out/r9j.err
  SUMMERSET_HIP_LIB=$V timeout 300 python bench.py --leg epaxos_cluster > gpurun_out/r9j_mul24_$i.json 2>> gpurun_out/r9j.err
done
SUMMERSET_HIP_LIB=$V timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "ep_ or epaxos or config4 or config5" 2>&1 | tail -4 > gpurun_out/r9j_tests_mul24.log
tail -2 gpurun_out/r9j_tests_mul24.log
python - <<P
import json
for i in (1, 2):
    for n in ("shipped", "mul24"):
        try:
            d = json.loads(open("gpurun_out/r9j_%s_%d.json" % (n, i)).read().strip().splitlines()[-1])
            d = d.get("epaxos_cluster", d)
            print(n, i, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if "ms_per_tick" in k or k in ("value",)})
        except Exception as e:
            print(n, i, "unreadable", e)
P
tail -3 gpurun_out/r9j.err
