#!/bin/bash
# r2s: wire ingest with the 8-bytes-at-a-time window reader (default build) against the byte reader (wi_coop0: also not cooperative), same call;
# then the kernel trace of the leg (how the call's time splits over the count pass, the scan and the write pass)
mkdir -p gpurun_out
R=$PWD
{ timeout 100 python -m pytest tests/test_zz_wire_ingest_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
V=$PWD/summerset_amd/variants
for lib in "" $V/libsummerset_hip_wi_coop0.so ""; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  echo "lib=$(basename "$lib")"; timeout 60 python bench.py --leg wire_ingest 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-420
done
unset SUMMERSET_HIP_LIB
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2s_prof -- python $R/bench.py --leg wire_ingest > /dev/null 2>&1
cd $R
DB=$(find gpurun_out/r2s_prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB | grep -i "kernel \|wire_ingest" | cut -c1-200
rm -rf gpurun_out/r2s_prof
} 2>&1 | tee gpurun_out/r2s_wi_reader.log
