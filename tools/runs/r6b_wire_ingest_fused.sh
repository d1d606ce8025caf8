#!/bin/bash
# smr_wire_ingest_mp as ONE launch (count, look-back, write; a tick-wide LDS window) against the two-launch build before it
# (variants/libsummerset_hip_wihead.so), interleaved; then the device tests of the parser on the new build
for rep in 1 2; do
  for v in head new; do
    if [ $v = head ]; then export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_wihead.so; else unset SUMMERSET_HIP_LIB; fi
    timeout 300 python bench.py --leg wire_ingest > gpurun_out/r6b_${v}_${rep}.json 2>> gpurun_out/r6b.err
    python - <<P
import json
d = json.loads(open("gpurun_out/r6b_${v}_${rep}.json").read().strip().splitlines()[-1])
print("$v $rep call_us", round(d["call_us"], 1), "frac", round(d["roofline"]["frac"], 3), "stream GB/s", round(d["stream_GBps"]))
P
  done
done
unset SUMMERSET_HIP_LIB
timeout 900 python -m pytest tests/test_zz_wire_ingest_gpu.py tests/test_zzz_wire_ingest_edges_gpu.py tests/test_zz_wire_emit_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
