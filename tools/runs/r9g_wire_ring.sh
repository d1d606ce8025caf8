#!/bin/bash
# wire ingest through a ring of two 128-byte lines, every line of the stream once per pass (csrc/wire_ingest.hip, round 5):
# its device tests, the leg against round 4's 128-byte window in the same call (summerset_amd/variants/libsummerset_hip_r4win.so),
# per-kernel times and the PMC passes of the leg
mkdir -p gpurun_out; R=$PWD
timeout 600 python -m pytest tests/test_zz_wire_ingest_gpu.py tests/test_zzz_wire_ingest_edges_gpu.py tests/test_zzz_example_raft_wire_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r9g_tests.log
tail -3 gpurun_out/r9g_tests.log
for k in 1 2; do
  timeout 200 python bench.py --leg wire_ingest > gpurun_out/r9g_leg_ring_$k.json 2>> gpurun_out/r9g.err
  SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_r4win.so timeout 200 python bench.py --leg wire_ingest > gpurun_out/r9g_leg_window_$k.json 2>> gpurun_out/r9g.err
done
( cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r9g_prof -- python $R/bench.py --leg wire_ingest > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r9g_pmc_fetch -- python $R/bench.py --leg wire_ingest > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r9g_pmc_write -- python $R/bench.py --leg wire_ingest > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/r9g_prof > gpurun_out/r9g_kernel_stats_wire_ingest_leg.txt 2>&1
python tools/pmc_traffic.py gpurun_out/r9g_pmc_fetch gpurun_out/r9g_pmc_write "bench.py --leg wire_ingest under rocprofv3 --pmc: the ring kernels" > gpurun_out/r9g_pmc_traffic_wire_ingest_leg.json 2>> gpurun_out/r9g.err
rm -rf gpurun_out/r9g_prof gpurun_out/r9g_pmc_fetch gpurun_out/r9g_pmc_write
grep "wire_ingest" gpurun_out/r9g_kernel_stats_wire_ingest_leg.txt | cut -c1-150
python - <<P
import json
for v in ("ring", "window"):
    for k in (1, 2):
        try:
            d = json.loads(open("gpurun_out/r9g_leg_%s_%d.json" % (v, k)).read().strip().splitlines()[-1])
            d = d.get("wire_ingest", d)
            print(v, k, "call us %.1f" % d["call_us"], "frac %.3f" % d["roofline"]["frac"])
        except Exception as e:
            print(v, k, "unreadable", e)
try:
    d = json.load(open("gpurun_out/r9g_pmc_traffic_wire_ingest_leg.json"))
    for k, v in d["kernels"].items():
        if "wire_ingest" in k: print("pmc", k, round(v["hbm_read_bytes_per_launch"] / 1e6, 1), "MB read", round(v["hbm_write_bytes_per_launch"] / 1e6, 1), "MB written")
except Exception as e:
    print("pmc unreadable", e)
P
tail -5 gpurun_out/r9g.err
