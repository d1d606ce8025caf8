#!/bin/bash
# r3j: smr_raft_leader_run_ticks: its parity tests on the device, the Raft leg (batches of 16 ticks in one launch beside one call per handler)
mkdir -p gpurun_out
{ timeout 600 python -m pytest tests/test_raft_gpu.py tests/test_zz_craft_gpu.py tests/test_zz_craft_follower_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python -c "
import json, torch, bench
print(json.dumps(bench.raft_leg(torch, torch.device('cuda:0'))))" 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r3j_leg_raft.json; cut -c1-2500 gpurun_out/r3j_leg_raft.json
} 2>&1 | tee gpurun_out/r3j.log
