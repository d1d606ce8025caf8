#!/bin/bash
# r3c: the one-launch EPaxos cluster tick on the device: its parity tests (one launch / launch by launch / driver loop / oracle
# cluster), the epaxos_cluster bench leg (both modes, same call), its kernel trace; and the kernel trace of the EXACT driver
# command merged over all of its processes (r3a's summary had picked a child process's database).
TAG=${1:-r3c}
mkdir -p gpurun_out
{ timeout 600 python -m pytest tests/test_zz_ep_cluster_gpu.py tests/test_zzz_ep_cluster_fused_gpu.py tests/test_zzz_example_ep_gpu.py tests/test_ep_gpu.py tests/test_zz_ep_exec_gpu.py tests/test_zz_ep_recovery_gpu.py tests/test_zzz_ep_recovery_exec_gpu.py tests/test_zzy_spread_ep_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5
timeout 300 python bench.py --leg epaxos_cluster 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${TAG}_leg_epaxos_cluster.json; cut -c1-1800 gpurun_out/${TAG}_leg_epaxos_cluster.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_epc -- python $R/bench.py --leg epaxos_cluster > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /dev/null
cd $R
python tools/rocpd_summary.py gpurun_out/${TAG}_prof_epc > gpurun_out/${TAG}_kernel_stats_epaxos_cluster.txt 2>&1
python tools/rocpd_summary.py gpurun_out/${TAG}_prof_bench --only mp_quorum_tally > gpurun_out/${TAG}_kernel_stats_default_bench.txt 2>&1
python tools/rocpd_summary.py gpurun_out/${TAG}_prof_bench > gpurun_out/${TAG}_kernel_stats_default_bench_all_processes.txt 2>&1
rm -rf gpurun_out/${TAG}_prof_bench gpurun_out/${TAG}_prof_epc
grep -v "at::native" gpurun_out/${TAG}_kernel_stats_epaxos_cluster.txt | head -8 | cut -c1-180
grep -v "at::native" gpurun_out/${TAG}_kernel_stats_default_bench.txt | head -14 | cut -c1-180
} 2>&1 | tee gpurun_out/${TAG}.log
