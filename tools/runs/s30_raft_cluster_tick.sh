# round 6: smr_raft_cluster_tick (append + replicate + replies of a co-located Raft / CRaft cluster in ONE launch) -- device tests, the craft_payload leg with it on / off (same call), kernel stats
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_raft_gpu.py tests/test_zz_craft_payload_gpu.py tests/test_zz_craft_gpu.py tests/test_zz_craft_follower_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -k "raft or craft or config2" -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/s30_tests.log; cat gpurun_out/s30_tests.log
for i in 1 2 3; do
  for on in 1 0; do
    SMR_RAFT_CLUSTER_TICK=$on timeout 300 python bench.py --leg craft_payload > gpurun_out/s30_leg_craft_payload_tick${on}_$i.json 2> gpurun_out/s30_leg_craft_payload_tick${on}_$i.err
    python - tick$on gpurun_out/s30_leg_craft_payload_tick${on}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms/tick %.4f" % d["ms_per_tick"], "engines %.4f" % d["engine_only_ms_per_tick"], "bytes path %.4f" % d.get("bytes_path_ms_per_tick", 0), "verified", d.get("verified"))
PY
  done
done
( cd /tmp && export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s30_prof -- python $GRAFT_REPO_ROOT/bench.py --leg craft_payload > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/s30_prof > gpurun_out/s30_kernel_stats_craft_payload_leg.txt 2>&1; rm -rf gpurun_out/s30_prof
grep "smr::" gpurun_out/s30_kernel_stats_craft_payload_leg.txt | head -10 | cut -c1-60,75-125
