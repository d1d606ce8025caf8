#!/bin/bash
# r2w: ep_pre_accept_replies_kernel with every input row loaded unconditionally from clamped addresses (default build) against
# the `on ? load : 0` form (ep_flat0), same call, through bench.py's epaxos leg; then the kernel's device parity tests
mkdir -p gpurun_out
{ V=$PWD/summerset_amd/variants
for lib in "" $V/libsummerset_hip_ep_flat0.so ""; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  echo "lib=$(basename "$lib")"
  timeout 40 python -c "import bench, torch, json; print(json.dumps(bench.epaxos_leg(torch, torch.device('cuda:0'))))" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-900
done
unset SUMMERSET_HIP_LIB
timeout 60 python -m pytest tests/test_ep_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r2w_ep_flat.log
