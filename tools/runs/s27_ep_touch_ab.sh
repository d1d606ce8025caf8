# round 6: ep_cluster_commit_one_by_one_kernel, the idle lanes of a wavefront asking for the cells the listed lanes' walks will read (EPC_CL_TOUCH) -- tests, then the leg A/B against -DEPC_CL_TOUCH=0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_zz_ep_cluster_gpu.py tests/test_zzz_ep_cluster_fused_gpu.py tests/test_zz_ep_exec_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -k "ep or config5" -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/s27_ep_tests.log; cat gpurun_out/s27_ep_tests.log
for i in 1 2 3; do
  for v in touch notouch; do
    if [ $v = notouch ]; then export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_notouch.so; else unset SUMMERSET_HIP_LIB; fi
    timeout 300 python bench.py --leg epaxos_cluster > gpurun_out/s27_leg_${v}_$i.json 2> gpurun_out/s27_leg_${v}_$i.err
    python - $v gpurun_out/s27_leg_${v}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
p = d["one_call_per_tick_phase_by_phase"]
print(sys.argv[1], "pm tick_us median %.1f min %.1f" % (p["tick_us_device_median"], p["tick_us_device_min"]), "same", p["same_commits_as_the_driver_loop"], p["same_commands_executed_as_the_driver_loop"])
PY
  done
done
unset SUMMERSET_HIP_LIB
( cd /tmp && export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s27_prof -- python $GRAFT_REPO_ROOT/bench.py --leg epaxos_cluster > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/s27_prof > gpurun_out/s27_kernel_stats_epaxos_leg.txt 2>&1; rm -rf gpurun_out/s27_prof
grep -i "ep_cluster\|commit_one" gpurun_out/s27_kernel_stats_epaxos_leg.txt | cut -c1-200
