#!/bin/bash
mkdir -p gpurun_out; R=$PWD
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r9f_prof -- python $R/bench.py --leg craft_payload > $R/gpurun_out/r9f_leg.json 2> /dev/null )
python tools/rocpd_summary.py gpurun_out/r9f_prof > gpurun_out/r9f_kernel_stats_craft_payload_leg.txt 2>&1
rm -rf gpurun_out/r9f_prof
grep -v "at::native" gpurun_out/r9f_kernel_stats_craft_payload_leg.txt | head -20 | cut -c1-160
