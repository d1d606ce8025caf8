#!/bin/bash
# round 5: what a tick of layout L2 is made of on one GPU (four virtual ranks): kernel trace of the l2 leg, steady state
mkdir -p gpurun_out
R=$PWD
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r8e_prof -- python $R/bench.py --leg l2 --steps 6 --timeouts 0 > $R/gpurun_out/r8e_l2.json 2> $R/gpurun_out/r8e.err )
python tools/rocpd_summary.py gpurun_out/r8e_prof > gpurun_out/r8e_kernel_stats.txt 2>&1
python tools/rocpd_timeline.py gpurun_out/r8e_prof "mp_,copy,Copy,fill" --limit 6000 > gpurun_out/r8e_timeline.txt 2>&1
rm -rf gpurun_out/r8e_prof
head -24 gpurun_out/r8e_kernel_stats.txt | cut -c1-170
tail -c 400 gpurun_out/r8e_l2.json
