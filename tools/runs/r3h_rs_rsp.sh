#!/bin/bash
# r3h: the one-pass from_data + encode kernel and the device-resident RSPaxos steady loop: their parity tests, the config-4 leg
mkdir -p gpurun_out
{ timeout 600 python -m pytest tests/test_rs_gpu.py tests/test_zz_rsp_steady_gpu.py tests/test_zz_rsp_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python bench.py --leg rspaxos 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r3h_leg_rspaxos.json; cut -c1-3000 gpurun_out/r3h_leg_rspaxos.json
} 2>&1 | tee gpurun_out/r3h.log
