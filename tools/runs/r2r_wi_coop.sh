#!/bin/bash
# r2r: wire ingest, refills loaded by the wavefront together (SMR_WI_COOP) against one lane per window, same call
mkdir -p gpurun_out
{ timeout 100 python -m pytest tests/test_zz_wire_ingest_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
V=$PWD/summerset_amd/variants
for lib in "" $V/libsummerset_hip_wi_coop0.so $V/libsummerset_hip_wi_coop1_256.so ""; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  echo "lib=$(basename "$lib")"; timeout 60 python bench.py --leg wire_ingest 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-420
done; } 2>&1 | tee gpurun_out/r2r_wi_coop.log
