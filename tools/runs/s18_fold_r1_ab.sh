# round 6: the next tick's appends in the tally launch -- device tests, then the driver's command with / without it, same call
# (when this ran the fold was on by default and SMR_MP_NO_FOLD_R1 turned it off; since then it is opt-in: SMR_MP_FOLD_R1)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mp_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -k "not ep and not raft and not rsp and not craft and not config2 and not config4 and not config5" -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/s18_mp_tests.log; cat gpurun_out/s18_mp_tests.log
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "ms/tick %.4f" % d["ms_per_step"], "tally us %.1f" % r["avg_launch_us"], "whole_tick frac_alg %.4f" % r["whole_tick"]["frac_alg"])
PY
}
for i in 1 2 3; do
  for v in fold nofold fold_defer; do
    unset SMR_MP_NO_FOLD_R1 SMR_MP_ALWAYS_DEFER_REST
    [ $v = nofold ] && export SMR_MP_NO_FOLD_R1=1
    [ $v = fold_defer ] && export SMR_MP_ALWAYS_DEFER_REST=1
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/s18_${v}_$i.json 2> gpurun_out/s18_${v}_$i.err
    line "$v driver" gpurun_out/s18_${v}_$i.json
  done
done
for v in fold nofold fold_defer; do
  unset SMR_MP_NO_FOLD_R1 SMR_MP_ALWAYS_DEFER_REST
  [ $v = nofold ] && export SMR_MP_NO_FOLD_R1=1
  [ $v = fold_defer ] && export SMR_MP_ALWAYS_DEFER_REST=1
  timeout 300 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra --no-l2 > gpurun_out/s18_${v}_steady.json 2> gpurun_out/s18_${v}_steady.err
  line "$v steady" gpurun_out/s18_${v}_steady.json
done
unset SMR_MP_NO_FOLD_R1 SMR_MP_ALWAYS_DEFER_REST
( cd /tmp && export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s18_prof -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra --no-l2 > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/s18_prof --only mp_straggler_batch > gpurun_out/s18_kernel_stats.txt 2>&1; rm -rf gpurun_out/s18_prof
head -14 gpurun_out/s18_kernel_stats.txt | cut -c1-160
