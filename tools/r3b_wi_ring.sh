#!/bin/bash
# r3b: wire ingest, the ring-of-lines variant (-DSMR_WI_RING=1) against the shipped parser, same call; its device parity tests first.
# Before the call, here:  tools/build_file_variant.sh wi_ring wire_ingest.hip -DSMR_WI_RING=1
mkdir -p gpurun_out
R=$PWD
V=$PWD/summerset_amd/variants
{ SUMMERSET_HIP_LIB=$V/libsummerset_hip_wi_ring.so timeout 150 python -m pytest tests/test_zz_wire_ingest_gpu.py tests/test_zzz_wire_ingest_edges_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
for lib in "" $V/libsummerset_hip_wi_ring.so "" $V/libsummerset_hip_wi_ring.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  echo "lib=$(basename "$lib")"; timeout 60 python bench.py --leg wire_ingest 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-700
done
export SUMMERSET_HIP_LIB=$V/libsummerset_hip_wi_ring.so
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3b_prof -- python $R/bench.py --leg wire_ingest > /dev/null 2>&1
cd $R
DB=$(find gpurun_out/r3b_prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB | grep -i "kernel \|wire_ingest" | cut -c1-200
rm -rf gpurun_out/r3b_prof
} 2>&1 | tee gpurun_out/r3b_wi_ring.log
