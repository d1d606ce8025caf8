#!/bin/bash
# The kernels written after round 1's GPU minutes were spent (ep_execute_kernel, rsp_*_kernel, the CRaft leader variant,
# qr_*_kernel) have run
# as host code only (tests/test_hostsim.py).  One gpurun call for their first device run:
#   gpurun --timeout 1800 -- 'bash tools/first_device_run.sh'
# 1. their device tests  2. their bench legs  3. a kernel trace of the legs (summary -> gpurun_out/)
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_zz_ep_exec_gpu.py tests/test_zz_mp_wide_gpu.py tests/test_zz_rsp_gpu.py tests/test_zz_rsp_bytes_gpu.py tests/test_zz_craft_gpu.py tests/test_zz_qread_gpu.py tests/test_zz_kv_gpu.py tests/test_mp_gpu.py::test_accept_replies_as_records -q -m gpu 2>&1 | tail -15 | tee gpurun_out/late_tests.log
for leg in epaxos_execution rspaxos_replica craft_leader quorum_read; do
    timeout 200 python bench.py --leg $leg > gpurun_out/leg_$leg.json 2> gpurun_out/leg_$leg.err || echo "leg $leg failed"
    tail -c 600 gpurun_out/leg_$leg.json
done
cd /tmp && export TMPDIR=/tmp
for leg in epaxos_execution rspaxos_replica craft_leader quorum_read; do
    timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$leg -- python $GRAFT_REPO_ROOT/bench.py --leg $leg > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py gpurun_out/prof_epaxos_execution > gpurun_out/prof_epaxos_execution.txt 2>&1
python tools/rocpd_summary.py gpurun_out/prof_rspaxos_replica > gpurun_out/prof_rspaxos_replica.txt 2>&1
python tools/rocpd_summary.py gpurun_out/prof_craft_leader > gpurun_out/prof_craft_leader.txt 2>&1
python tools/rocpd_summary.py gpurun_out/prof_quorum_read > gpurun_out/prof_quorum_read.txt 2>&1
