#!/bin/bash
mkdir -p gpurun_out
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3i_prof -- python $R/bench.py --leg rspaxos > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/r3i_prof > gpurun_out/r3i_kernel_stats_rspaxos_leg.txt 2>&1
rm -rf gpurun_out/r3i_prof
head -30 gpurun_out/r3i_kernel_stats_rspaxos_leg.txt | cut -c1-175
