#!/bin/bash
# the ingest kernels with the next window's lines touched a round ahead (-DSMR_WI_TOUCH=1: the one-pass kernel; =2: also the counting pass of the two-pass call), same call x4
mkdir -p gpurun_out; R=$PWD
for v in touch touch2; do SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_$v.so timeout 600 python -m pytest tests/test_zz_wire_ingest_conn_gpu.py tests/test_zz_wire_ingest_gpu.py tests/test_zzz_wire_ingest_edges_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
for k in 1 2 3 4; do
  for v in shipped touch touch2; do
    L=$R/summerset_amd/libsummerset_hip.so; [ $v != shipped ] && L=$R/summerset_amd/variants/libsummerset_hip_$v.so
    SUMMERSET_HIP_LIB=$L timeout 200 python bench.py --leg wire_ingest > gpurun_out/r9o_${v}_$k.json 2>> gpurun_out/r9o.err
  done
done
python - <<P
import json
for k in (1, 2, 3, 4):
    for v in ("shipped", "touch", "touch2"):
        d = json.loads(open("gpurun_out/r9o_%s_%d.json" % (v, k)).read().strip().splitlines()[-1])
        print(v, k, "one pass: call us %.1f frac %.3f | dense lists: call us %.1f" % (d["call_us"], d["roofline"]["frac"], d["dense_lists"]["call_us"]))
P
tail -3 gpurun_out/r9o.err
