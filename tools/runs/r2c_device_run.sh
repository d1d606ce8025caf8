#!/bin/bash
# Round 2, after adopting the five A/B'd variants as the default build: the whole -m gpu suite (incl. the new per-BASELINE-
# config slice tests), the bench line, then the rocprofv3 passes over tools/pmc_probe.py on the DEFAULT workload.
TAG=${1:-r2c}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 2>&1 | tail -60 > gpurun_out/${TAG}_gputests.log
tail -3 gpurun_out/${TAG}_gputests.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 1500 gpurun_out/${TAG}_bench.json
timeout 300 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra > gpurun_out/${TAG}_bench_steady.json 2>> gpurun_out/${TAG}_bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -- python $R/tools/pmc_probe.py --extra > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python $R/tools/pmc_probe.py --extra > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python $R/tools/pmc_probe.py --extra > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_prof -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats.txt 2>&1
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write "tools/pmc_probe.py --extra: 16 ticks of the bench shape on the default workload through the per-round kernels, 32 more through the fused tick kernel (2 launches of 16) (65536 groups x 5, S=32, H=4, 10% loss, 1% leader changes) + 3 RS(3,2) encodes of 65536 x 4099 B + the Raft / EPaxos legs" > gpurun_out/${TAG}_pmc_traffic.json 2> gpurun_out/${TAG}_pmc_traffic.err
head -30 gpurun_out/${TAG}_kernel_stats.txt
