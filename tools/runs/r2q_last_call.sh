#!/bin/bash
# r2q: the round's last GPU minutes -- first device run of the spread EPaxos layout, the wire ingest at its new window
mkdir -p gpurun_out
{ timeout 150 python -m pytest tests/test_zzy_spread_ep_gpu.py tests/test_zz_wire_ingest_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -4
  timeout 60 python bench.py --leg wire_ingest 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 60 python bench.py --layout spread-epaxos --groups 65536 --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1
} 2>&1 | tee gpurun_out/r2q_last_call.log
