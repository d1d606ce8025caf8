mkdir -p gpurun_out
TAG=${1:-s1}
timeout 600 python -m pytest tests/test_zz_ep_cluster_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q -k "ep or config4 or config5 or epaxos" > gpurun_out/${TAG}_ep_tests.log 2>&1; tail -3 gpurun_out/${TAG}_ep_tests.log
timeout 400 python bench.py --leg epaxos_cluster > gpurun_out/${TAG}_leg_epaxos_cluster.json 2> gpurun_out/${TAG}_leg_epaxos_cluster.err
python - <<P
import json
d=json.loads(open("gpurun_out/${TAG}_leg_epaxos_cluster.json").read().strip().splitlines()[-1])
for k in ("one_call_per_tick","one_call_per_tick_phase_by_phase"):
    v=d[k]; print(k, {a:b for a,b in v.items() if a in ("ms_per_tick","tick_us_device_median","tick_us_device_min","same_commits_as_the_driver_loop","same_commands_executed_as_the_driver_loop","batch_stats","error")})
P
SMR_EP_PM_UNBATCHED=1 timeout 400 python bench.py --leg epaxos_cluster 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['one_call_per_tick_phase_by_phase']; print('unbatched A/B:', v.get('tick_us_device_median'), v.get('tick_us_device_min'))"
