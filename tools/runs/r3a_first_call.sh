#!/bin/bash
# Round 3, first device call: the record of the SHIPPED build.
#   whole -m gpu suite (no -x, no xfail marks left) | default bench line | driver flags | steady
#   rocprofv3 --kernel-trace --stats over the EXACT driver command (bench.py --gpus 1 --steps 20 --warmup 5)
#   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) over tools/pmc_probe.py --extra
#   kernel trace of the EPaxos cluster leg; the wire-ingest ring-of-lines A/B (tools/runs/r3b_wi_ring.sh)
TAG=${1:-r3a}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rxX -p no:cacheprovider --durations=8 2>&1 | tail -60 > gpurun_out/${TAG}_gputests.log
tail -3 gpurun_out/${TAG}_gputests.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 400 gpurun_out/${TAG}_bench.json; echo
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra > gpurun_out/${TAG}_bench_driver_flags.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra > gpurun_out/${TAG}_bench_steady.json 2>> gpurun_out/${TAG}_bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /dev/null
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_epc -- python $R/bench.py --leg epaxos_cluster > $R/gpurun_out/${TAG}_leg_epaxos_cluster.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python $R/tools/pmc_probe.py --extra > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python $R/tools/pmc_probe.py --extra > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_prof_bench -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats_default_bench.txt 2>&1
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_prof_epc -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats_epaxos_cluster.txt 2>&1
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write "tools/pmc_probe.py --extra at HEAD: 16 ticks of the bench shape on the default workload (65536 groups x 5, S=32, H=4, 10% loss, 1% leader changes) as bench.py runs them (two smr_mp_run_ticks batches of 8, straggler list on, ttl 4), 32 more through the fused tick kernel + 3 RS(3,2) encodes of 65536 x 4099 B + the Raft / EPaxos / wire-ingest legs" > gpurun_out/${TAG}_pmc_traffic.json 2> gpurun_out/${TAG}_pmc_traffic.err
rm -rf gpurun_out/${TAG}_prof_bench gpurun_out/${TAG}_prof_epc gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write
grep -v "at::native" gpurun_out/${TAG}_kernel_stats_default_bench.txt | head -12 | cut -c1-150
grep -v "at::native" gpurun_out/${TAG}_kernel_stats_epaxos_cluster.txt | head -8 | cut -c1-150
[ -f summerset_amd/variants/libsummerset_hip_wi_ring.so ] && bash tools/runs/r3b_wi_ring.sh > /dev/null 2>&1; tail -12 gpurun_out/r3b_wi_ring.log | cut -c1-300
