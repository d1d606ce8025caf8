#!/bin/bash
# r2p: the device-side peer-traffic ingest (csrc/wire_ingest.hip): its device tests, then the bench leg at three LDS window sizes
mkdir -p gpurun_out
{ timeout 600 python -m pytest tests/test_zz_wire_ingest_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
V=$PWD/summerset_amd/variants
for lib in "" $V/libsummerset_hip_wi128.so $V/libsummerset_hip_wi512.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  echo "lib=$(basename "$lib")"; timeout 200 python bench.py --leg wire_ingest 2>&1 | grep -v amdgpu.ids | tail -1
done; } 2>&1 | tee gpurun_out/r2p_wire_ingest.log
