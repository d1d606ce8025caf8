mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_zz_ep_cluster_gpu.py tests/test_zzz_ep_cluster_fused_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -k "ep or config5 or config4" -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/s17_ep_tests.log; cat gpurun_out/s17_ep_tests.log
for i in 1 2; do
  for v in wide ephead; do
    if [ $v = ephead ]; then export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_ephead.so; else unset SUMMERSET_HIP_LIB; fi
    timeout 300 python bench.py --leg epaxos_cluster > gpurun_out/s17_leg_${v}_$i.json 2> gpurun_out/s17_leg_${v}_$i.err
    python - $v gpurun_out/s17_leg_${v}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
p = d["one_call_per_tick_phase_by_phase"]
print(sys.argv[1], "pm tick_us median %.1f min %.1f" % (p["tick_us_device_median"], p["tick_us_device_min"]), "loops order %.1f" % d["one_call_per_tick"]["tick_us_device_median"])
PY
  done
done
unset SUMMERSET_HIP_LIB
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/s17_headline.json 2> gpurun_out/s17_headline.err; tail -c 600 gpurun_out/s17_headline.json
