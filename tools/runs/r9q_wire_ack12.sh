#!/bin/bash
# the one-pass ingest with 12-byte (slot, ballot) records in the segments (smr_wire_ack12): device tests, the leg, kernel stats, PMC
mkdir -p gpurun_out; R=$PWD
timeout 600 python -m pytest tests/test_zz_wire_ingest_conn_gpu.py tests/test_zz_wire_ingest_gpu.py tests/test_zzz_wire_ingest_edges_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r9q_tests.log
tail -3 gpurun_out/r9q_tests.log
for k in 1 2; do
  timeout 200 python bench.py --leg wire_ingest > gpurun_out/r9q_leg_wire_ingest_$k.json 2>> gpurun_out/r9q.err
done
( cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r9q_prof -- python $R/bench.py --leg wire_ingest > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r9q_pmc_fetch -- python $R/bench.py --leg wire_ingest > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r9q_pmc_write -- python $R/bench.py --leg wire_ingest > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/r9q_prof > gpurun_out/r9q_kernel_stats_wire_ingest_leg.txt 2>&1
python tools/pmc_traffic.py gpurun_out/r9q_pmc_fetch gpurun_out/r9q_pmc_write "bench.py --leg wire_ingest under rocprofv3 --pmc: the two-pass call and the one-pass call" > gpurun_out/r9q_pmc_traffic_wire_ingest_leg.json 2>> gpurun_out/r9q.err
rm -rf gpurun_out/r9q_prof gpurun_out/r9q_pmc_fetch gpurun_out/r9q_pmc_write
grep "wire_ingest" gpurun_out/r9q_kernel_stats_wire_ingest_leg.txt | cut -c1-150
python - <<P
import json
for k in (1, 2):
    try:
        d = json.loads(open("gpurun_out/r9q_leg_wire_ingest_%d.json" % k).read().strip().splitlines()[-1])
        print(k, "one pass: call us %.1f frac %.3f | dense lists: call us %.1f frac %.3f" % (d["call_us"], d["roofline"]["frac"], d["dense_lists"]["call_us"], d["dense_lists"]["roofline"]["frac"]))
    except Exception as e:
        print(k, "unreadable", e)
try:
    d = json.load(open("gpurun_out/r9q_pmc_traffic_wire_ingest_leg.json"))
    for k, v in d["kernels"].items():
        if "wire_ingest" in k: print("pmc", k, round(v["hbm_read_bytes_per_launch"] / 1e6, 1), "MB read", round(v["hbm_write_bytes_per_launch"] / 1e6, 1), "MB written")
except Exception as e:
    print("pmc unreadable", e)
P
tail -5 gpurun_out/r9q.err
