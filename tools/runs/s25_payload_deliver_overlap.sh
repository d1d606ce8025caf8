# round 6: ps_put_deliver_kernel with the batch loads issued ahead of the decision barrier -- same run as s24
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zz_rsp_payload_gpu.py tests/test_zz_craft_payload_gpu.py tests/test_zz_craft_gpu.py tests/test_zzz_example_rsp_payload_gpu.py tests/test_zzzz_rsp_emit_accepts_gpu.py tests/test_zz_rsp_bytes_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -k "payload or craft or rsp or config3 or config4" -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/s25_payload_tests.log; cat gpurun_out/s25_payload_tests.log
for leg in rspaxos_payload craft_payload; do
  for i in 1 2; do
    for dl in 1 0; do
      SMR_PS_DELIVER=$dl timeout 300 python bench.py --leg $leg > gpurun_out/s25_leg_${leg}_dl${dl}_$i.json 2> gpurun_out/s25_leg_${leg}_dl${dl}_$i.err
      python - $leg$dl gpurun_out/s25_leg_${leg}_dl${dl}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms/tick %.4f" % d["ms_per_tick"], "bytes path %.4f" % d.get("bytes_path_ms_per_tick", 0), "verified", d.get("verified"))
PY
    done
  done
  ( cd /tmp && export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s25_prof -- python $GRAFT_REPO_ROOT/bench.py --leg $leg > /dev/null 2>&1 )
  python tools/rocpd_summary.py gpurun_out/s25_prof > gpurun_out/s25_kernel_stats_${leg}_leg.txt 2>&1; rm -rf gpurun_out/s25_prof
  grep "smr::" gpurun_out/s25_kernel_stats_${leg}_leg.txt | head -10 | cut -c1-60,75-125
done
