#!/bin/bash
# mp_round_local's register budget (shipped: R1_PF 32 tokens prefetched, 158 VGPRs, 3 wavefronts per SIMD): R1_PF 16 / 8 and launch bounds 4 / 5, driver's command + steady, same call
mkdir -p gpurun_out; R=$PWD
for i in 1 2; do
  for v in shipped pf16 pf16w4 pf32w4 pf8w5; do
    L=$R/summerset_amd/libsummerset_hip.so; [ $v != shipped ] && L=$R/summerset_amd/variants/libsummerset_hip_$v.so
    SUMMERSET_HIP_LIB=$L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r9m_${v}_$i.json 2>> gpurun_out/r9m.err
    cp bench_detail.json gpurun_out/r9m_${v}_detail_$i.json
    SUMMERSET_HIP_LIB=$L timeout 300 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r9m_${v}_steady_$i.json 2>> gpurun_out/r9m.err
    cp bench_detail.json gpurun_out/r9m_${v}_steady_detail_$i.json
  done
done
python - <<P
import json
def r1(f):
    try:
        k = json.load(open(f)).get("kernels") or {}
        for n, v in k.items():
            if "R1" in n or "local" in n: return v if not isinstance(v, dict) else v.get("avg_us", v)
    except Exception as e:
        return None
for i in (1, 2):
    for n in ("shipped", "pf16", "pf16w4", "pf32w4", "pf8w5"):
        try:
            d = json.loads(open("gpurun_out/r9m_%s_%d.json" % (n, i)).read().strip().splitlines()[-1])
            s = json.loads(open("gpurun_out/r9m_%s_steady_%d.json" % (n, i)).read().strip().splitlines()[-1])
            print(n, i, "ms/tick %.4f | steady %.4f" % (d["ms_per_step"], s["ms_per_step"]), "R1", r1("gpurun_out/r9m_%s_detail_%d.json" % (n, i)), r1("gpurun_out/r9m_%s_steady_detail_%d.json" % (n, i)))
        except Exception as e:
            print(n, i, "unreadable", e)
P
tail -3 gpurun_out/r9m.err
