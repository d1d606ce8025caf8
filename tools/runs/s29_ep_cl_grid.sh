# round 6: ep_cluster_commit_one_by_one_kernel with the replica as blockIdx.x, empty blocks leaving at once, 2 listed lanes per wavefront (new) against the old grid at 2 and 1 lanes (variants cl2, cl1 built from the tree before)
mkdir -p gpurun_out
for i in 1 2; do
  for v in new cl2 cl1; do
    if [ $v = new ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$v.so; fi
    timeout 300 python bench.py --leg epaxos_cluster > gpurun_out/s29_leg_${v}_$i.json 2> gpurun_out/s29_leg_${v}_$i.err
    python - $v gpurun_out/s29_leg_${v}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
p = d["one_call_per_tick_phase_by_phase"]
print(sys.argv[1], "pm tick_us median %.1f min %.1f" % (p["tick_us_device_median"], p["tick_us_device_min"]), "same", p["same_commits_as_the_driver_loop"], p["same_commands_executed_as_the_driver_loop"])
PY
  done
done
unset SUMMERSET_HIP_LIB
timeout 600 python -m pytest tests/test_zz_ep_cluster_gpu.py tests/test_zzz_ep_cluster_fused_gpu.py tests/test_zz_ep_exec_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -k "ep or config5" -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/s29_ep_tests.log; cat gpurun_out/s29_ep_tests.log
( cd /tmp && export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s29_prof -- python $GRAFT_REPO_ROOT/bench.py --leg epaxos_cluster > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/s29_prof > gpurun_out/s29_kernel_stats_epaxos_leg.txt 2>&1; rm -rf gpurun_out/s29_prof
grep -i "ep_cluster\|commit_one" gpurun_out/s29_kernel_stats_epaxos_leg.txt | cut -c1-200
