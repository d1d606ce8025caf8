#!/bin/bash
# (1) same-call A/B: the shipped tally (4 rows per pass, 82 VGPRs, 5 wavefronts per SIMD) against -DTALLY_MINW=6 (80 VGPRs, 12 B scratch, 6)
# (2) the RSPaxos leg with every shard written once, its tests, the RS tests
R=$PWD
B="--no-l2 --no-extra --no-rs --no-cpu"
for i in 1 2; do
  for v in new tw6; do
    if [ $v = new ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_$v.so; fi
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r5f_${v}_driver_$i.json 2>> gpurun_out/r5f.err
    timeout 200 python bench.py --timeouts 0 $B > gpurun_out/r5f_${v}_steady_$i.json 2>> gpurun_out/r5f.err
  done
done
unset SUMMERSET_HIP_LIB
python - <<P
import json
for v in ("new", "tw6"):
    for k in ("driver", "steady"):
        for i in (1, 2):
            try:
                d = json.loads(open("gpurun_out/r5f_%s_%s_%d.json" % (v, k, i)).read().strip().splitlines()[-1])
                ks = d["kernels"]
                print(v, k, i, "ms/tick %.4f" % d["ms_per_step"], " ".join("%s %.1f" % (n.split("_")[0], ks[n]["avg_us"]) for n in ks), "tally frac %.3f" % d["roofline"]["frac"])
            except Exception as e:
                print(v, k, i, "unreadable", e)
P
timeout 300 python bench.py --leg rspaxos > gpurun_out/r5f_leg_rspaxos.json 2>> gpurun_out/r5f.err
python - <<P
import json
d = json.loads(open("gpurun_out/r5f_leg_rspaxos.json").read().strip().splitlines()[-1])
print("rspaxos leg: value %.4g ms/tick %.4f" % (d["value"], d["ms_per_tick"]), "graph", d.get("graph", {}).get("device_ms_per_tick"), "eager", d["eager"]["device_ms_per_tick"])
print("  roofline", {k: d["roofline"][k] for k in ("frac", "frac_on_survey_8d_bytes", "avg_launch_us")} if "roofline" in d else None)
print("  encode", d.get("from_data_and_encode"))
P
timeout 900 python -m pytest tests/test_rs_gpu.py tests/test_zz_rsp_steady_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q -x -p no:cacheprovider -k "rs or rsp or config3" 2>&1 | tail -3
