#!/bin/bash
# round 5: R2's bulk launch split into the fast path alone (79 VGPRs, no scratch) + the rest (its own launch) beside a busy side stream:
# the MultiPaxos device tests, then the driver's command with the split and, same call, without (SMR_MP_NO_SPLIT_R2=1), twice
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mp_gpu.py tests/test_zz_mp_wide_gpu.py tests/test_baseline_configs_gpu.py tests/test_example_gpu.py -k "not config3 and not config4 and not config5 and not payload" -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r8i_tests.log
tail -3 gpurun_out/r8i_tests.log
for i in 1 2; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8i_split_$i.json 2>> gpurun_out/r8i.err
  SMR_MP_NO_SPLIT_R2=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8i_nosplit_$i.json 2>> gpurun_out/r8i.err
done
timeout 300 python bench.py --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8i_split_default.json 2>> gpurun_out/r8i.err
SMR_MP_NO_SPLIT_R2=1 timeout 300 python bench.py --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/r8i_nosplit_default.json 2>> gpurun_out/r8i.err
python - <<P
import json
for n in ("split_1", "nosplit_1", "split_2", "nosplit_2", "split_default", "nosplit_default"):
    d = json.loads(open("gpurun_out/r8i_%s.json" % n).read().strip().splitlines()[-1])
    k = json.load(open("gpurun_out/bench_detail.json"))["kernels"] if False else None
    print(n, "ms/tick %.4f  tally us %.1f frac %.3f whole-tick frac_alg %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["whole_tick"]["frac_alg"]))
P
R=$PWD
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r8i_prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/r8i_prof --only mp_quorum_tally > gpurun_out/r8i_kernel_stats.txt 2>&1
rm -rf gpurun_out/r8i_prof
head -12 gpurun_out/r8i_kernel_stats.txt | cut -c1-150
