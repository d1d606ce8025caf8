#!/bin/bash
# r4c: the Raft / EPaxos reply parsers on the device: their tests, the reply_ingest leg, its kernel times; the MultiPaxos ingest suite again
TAG=${1:-r4c}
mkdir -p gpurun_out
R=$PWD; export PYTHONPATH=$R
{ timeout 900 python -m pytest tests/test_zz_reply_ingest_gpu.py tests/test_zz_wire_ingest_gpu.py tests/test_zzz_wire_ingest_edges_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python bench.py --leg reply_ingest 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${TAG}_leg_reply_ingest.json; cut -c1-900 gpurun_out/${TAG}_leg_reply_ingest.json; echo
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ri -o ri -- python $R/bench.py --leg reply_ingest > /tmp/prof_ri.log 2>&1 )
python tools/rocpd_summary.py /tmp/prof_ri --only wire_ingest > gpurun_out/${TAG}_kernel_stats_reply_ingest.txt 2>&1; cut -c1-200 gpurun_out/${TAG}_kernel_stats_reply_ingest.txt
} 2>&1 | tee gpurun_out/${TAG}.log
