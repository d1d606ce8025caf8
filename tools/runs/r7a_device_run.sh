mkdir -p gpurun_out
T=r7a
timeout 300 python -m pytest tests/test_zz_rsp_payload_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/${T}_payload_tests.log; tail -3 gpurun_out/${T}_payload_tests.log
timeout 200 python bench.py --leg rspaxos_payload > gpurun_out/${T}_leg_rspaxos_payload.json 2> gpurun_out/${T}_leg.err; tail -c 900 gpurun_out/${T}_leg_rspaxos_payload.json; tail -3 gpurun_out/${T}_leg.err
timeout 400 python -m pytest tests -m gpu -q -rxX -p no:cacheprovider --durations=6 2>&1 | tail -25 > gpurun_out/${T}_gputests.log; tail -3 gpurun_out/${T}_gputests.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_command.json 2> gpurun_out/${T}_bench.err; python - <<P
import json
d = json.loads(open("gpurun_out/${T}_bench_driver_command.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.4f tally frac %.3f legs_failed %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("legs_failed")))
print("rspaxos_payload", json.dumps(d.get("rspaxos_payload"))[:300])
P
( cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
  timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof_payload -- python $R/bench.py --leg rspaxos_payload > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/${T}_prof_payload > gpurun_out/${T}_kernel_stats_payload_leg.txt 2>&1; rm -rf gpurun_out/${T}_prof_payload; head -14 gpurun_out/${T}_kernel_stats_payload_leg.txt | cut -c1-150
