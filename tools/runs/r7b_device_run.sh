# r7b, r7c: the payload store after the Horner rebuild / forwarding byte kernel / no memset launch: its device tests, the leg, kernel stats
mkdir -p gpurun_out
T=r7e
timeout 300 python -m pytest tests/test_zz_rsp_payload_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/${T}_payload_tests.log; tail -3 gpurun_out/${T}_payload_tests.log
timeout 200 python bench.py --leg rspaxos_payload > gpurun_out/${T}_leg_rspaxos_payload.json 2> gpurun_out/${T}_leg.err; tail -c 700 gpurun_out/${T}_leg_rspaxos_payload.json; tail -3 gpurun_out/${T}_leg.err
( cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
  timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof_payload -- python $R/bench.py --leg rspaxos_payload > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/${T}_prof_payload > gpurun_out/${T}_kernel_stats_payload_leg.txt 2>&1; rm -rf gpurun_out/${T}_prof_payload; head -12 gpurun_out/${T}_kernel_stats_payload_leg.txt | cut -c1-150
