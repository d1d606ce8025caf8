#!/bin/bash
# round 5: two ranks of the library's communicator on ONE device -- does this RCCL take them?
mkdir -p gpurun_out
export NCCL_DEBUG=WARN
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/two_ranks_one_gpu.py > gpurun_out/r8g_two_ranks_one_gpu.log 2>&1
echo "rc $?"; grep -v "^W0\|^\*\*\*\|OMP_NUM" gpurun_out/r8g_two_ranks_one_gpu.log | tail -12
