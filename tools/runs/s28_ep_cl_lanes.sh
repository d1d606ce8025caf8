# round 6: ep_cluster_commit_one_by_one_kernel with the replica as blockIdx.x, empty blocks leaving at once, 2 listed lanes per wavefront (new) against the old grid at 2 and 1 lanes (variants cl2, cl1 built from the tree before)
mkdir -p gpurun_out
for i in 1 2; do
  for v in cl4 cl2 cl1 cl8; do
    if [ $v = cl4 ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$v.so; fi
    timeout 300 python bench.py --leg epaxos_cluster > gpurun_out/s28_leg_${v}_$i.json 2> gpurun_out/s28_leg_${v}_$i.err
    python - $v gpurun_out/s28_leg_${v}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
p = d["one_call_per_tick_phase_by_phase"]
print(sys.argv[1], "pm tick_us median %.1f min %.1f" % (p["tick_us_device_median"], p["tick_us_device_min"]), "same", p["same_commits_as_the_driver_loop"], p["same_commands_executed_as_the_driver_loop"])
PY
  done
done
