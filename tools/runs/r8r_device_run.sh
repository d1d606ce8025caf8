#!/bin/bash
# round 5: layout L2 with the rounds of a rank's blocks in ONE launch (and one multi-tensor copy per virtual exchange): its device
# tests, the l2 leg (four virtual ranks on one GPU), a kernel trace of it
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_spread_mp.py tests/test_comm.py tests/test_zzz_spread_qread_device_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/r8r_tests.log
tail -2 gpurun_out/r8r_tests.log
timeout 300 python bench.py --leg l2 --steps 12 > gpurun_out/r8r_l2.json 2> gpurun_out/r8r.err
python - <<P
import json
d = json.loads(open("gpurun_out/r8r_l2.json").read().strip().splitlines()[-1])
print("l2 ms/tick %.3f (per virtual rank %.3f) steady %.3f (per virtual rank %.3f)" % (d["ms_per_step"], d["ms_per_step"] / 4, d["steady_state"]["ms_per_tick"], d["steady_state"]["ms_per_tick"] / 4))
P
R=$PWD
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r8r_prof -- python $R/bench.py --leg l2 --steps 6 --timeouts 0 > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/r8r_prof > gpurun_out/r8r_kernel_stats.txt 2>&1
rm -rf gpurun_out/r8r_prof
head -14 gpurun_out/r8r_kernel_stats.txt | cut -c1-150
