#!/bin/bash
# round 5: where do mp_round_deliver's long launches come from?  kernel trace of the driver's command (headline only), per launch
mkdir -p gpurun_out
R=$PWD
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r8c_prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > $R/gpurun_out/r8c_bench.json 2> $R/gpurun_out/r8c_bench.err )
python tools/rocpd_timeline.py gpurun_out/r8c_prof "mp_" --only mp_quorum_tally --limit 1400 > gpurun_out/r8c_timeline.txt 2>&1
python tools/rocpd_summary.py gpurun_out/r8c_prof --only mp_quorum_tally > gpurun_out/r8c_kernel_stats.txt 2>&1
rm -rf gpurun_out/r8c_prof
tail -c 600 gpurun_out/r8c_bench.json; echo; head -12 gpurun_out/r8c_kernel_stats.txt | cut -c1-160
