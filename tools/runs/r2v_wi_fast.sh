#!/bin/bash
# r2v: wire ingest with short payloads decoded out of two registers (Reg128) + the scan kernel's batched loads (default build)
# against the build before them (wi_rd8), same call; kernel trace of the default
mkdir -p gpurun_out
R=$PWD
{ timeout 100 python -m pytest tests/test_zz_wire_ingest_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
V=$PWD/summerset_amd/variants
for lib in "" $V/libsummerset_hip_wi_rd8.so ""; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  echo "lib=$(basename "$lib")"; timeout 60 python bench.py --leg wire_ingest 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-700
done
unset SUMMERSET_HIP_LIB
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2v_prof -- python $R/bench.py --leg wire_ingest > /dev/null 2>&1
cd $R
DB=$(find gpurun_out/r2v_prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB | grep -i "kernel \|wire_ingest" | cut -c1-200
rm -rf gpurun_out/r2v_prof
} 2>&1 | tee gpurun_out/r2v_wi_fast.log
