#!/bin/bash
# round 5: the side stream's batch kernel with every round of a listed group as a wave-cooperative job (STRAG_COOP):
# the MultiPaxos device tests, the driver's command (x2) + default + steady lines, kernel trace + timeline of the driver's command
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mp_gpu.py tests/test_zz_mp_wide_gpu.py tests/test_baseline_configs_gpu.py tests/test_example_gpu.py -k "not config3 and not config4 and not config5 and not payload" -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r8j_tests.log
tail -3 gpurun_out/r8j_tests.log
for i in 1 2; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8j_driver_$i.json 2>> gpurun_out/r8j.err
done
timeout 300 python bench.py --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8j_default.json 2>> gpurun_out/r8j.err
timeout 300 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8j_steady.json 2>> gpurun_out/r8j.err
python - <<P
import json
for n in ("driver_1", "driver_2", "default", "steady"):
    d = json.loads(open("gpurun_out/r8j_%s.json" % n).read().strip().splitlines()[-1])
    print(n, "ms/tick %.4f  tally us %.1f frac %.3f whole-tick frac_alg %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["whole_tick"]["frac_alg"]))
P
R=$PWD
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r8j_prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/r8j_prof --only mp_quorum_tally > gpurun_out/r8j_kernel_stats.txt 2>&1
python tools/rocpd_timeline.py gpurun_out/r8j_prof "mp_" --only mp_quorum_tally --limit 400 > gpurun_out/r8j_timeline.txt 2>&1
rm -rf gpurun_out/r8j_prof
head -11 gpurun_out/r8j_kernel_stats.txt | cut -c1-150
