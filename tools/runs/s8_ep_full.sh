TAG=${1:-s8}
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q -k "ep_ or epaxos or config4 or config5 or spread_ep" > gpurun_out/${TAG}_ep_tests.log 2>&1; tail -3 gpurun_out/${TAG}_ep_tests.log
( cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -- python $R/bench.py --leg epaxos_cluster > $R/gpurun_out/${TAG}_leg_epaxos_cluster.json 2> /dev/null
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python $R/bench.py --leg epaxos_cluster > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python $R/bench.py --leg epaxos_cluster > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/${TAG}_prof > gpurun_out/${TAG}_kernel_stats_epaxos_leg.txt 2>&1
grep -i "ep_cluster\|commit_one" gpurun_out/${TAG}_kernel_stats_epaxos_leg.txt | cut -c1-220
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write "bench.py --leg epaxos_cluster under rocprofv3 --pmc" > gpurun_out/${TAG}_pmc_traffic_epaxos_leg.json 2> gpurun_out/${TAG}_pmc.err
python - <<P
import json
d=json.load(open("gpurun_out/${TAG}_pmc_traffic_epaxos_leg.json"))
for k,v in d["kernels"].items():
    if "ep_cluster" in k: print("pmc", k[:70], v["launches"], round(v["hbm_read_bytes_per_launch"]/1e6,1), "+", round(v["hbm_write_bytes_per_launch"]/1e6,1), "MB")
d=json.loads(open("gpurun_out/${TAG}_leg_epaxos_cluster.json").read().strip().splitlines()[-1])
for k in ("one_call_per_tick","one_call_per_tick_phase_by_phase"):
    v=d[k]; print(k, {a:b for a,b in v.items() if a in ("ms_per_tick","tick_us_device_median","tick_us_device_min","same_commits_as_the_driver_loop","same_commands_executed_as_the_driver_loop","batch_stats","error")})
P
rm -rf gpurun_out/${TAG}_prof gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write
