#!/bin/bash
# r3w: wire ingest after a change: its parity tests on the device, the leg, per-kernel times (rocprofv3 --kernel-trace), then counters
# (instruction mix / waits, LDS, FETCH_SIZE and WRITE_SIZE in separate passes) of the leg's kernels.   usage: r3w_wi.sh TAG [nopmc]
TAG=${1:-r3w}
mkdir -p gpurun_out
R=$PWD; export PYTHONPATH=$R
cat > gpurun_out/${TAG}_wi_leg.py <<'P'
import json, torch, bench
print(json.dumps(bench.wire_ingest_leg(torch, torch.device('cuda:0'), iters=8)))
P
{ timeout 900 python -m pytest tests/test_zz_wire_ingest_gpu.py tests/test_zzz_wire_ingest_edges_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python gpurun_out/${TAG}_wi_leg.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${TAG}_leg_wire_ingest.json; cut -c1-900 gpurun_out/${TAG}_leg_wire_ingest.json; echo
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_wi -o wi -- python $R/gpurun_out/${TAG}_wi_leg.py > /tmp/prof_wi.log 2>&1 )
python tools/rocpd_summary.py /tmp/prof_wi --only wire_ingest > gpurun_out/${TAG}_kernel_stats_wire_ingest.txt 2>&1; cut -c1-200 gpurun_out/${TAG}_kernel_stats_wire_ingest.txt
} 2>&1 | tee gpurun_out/${TAG}.log
[ "$2" = nopmc ] && exit 0
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_$tag -- python $R/gpurun_out/${TAG}_wi_leg.py > /tmp/pmc_$tag.log 2>&1 || tail -3 /tmp/pmc_$tag.log
done
cd $R
python - $TAG <<'PY' | tee gpurun_out/${TAG}_wi_pmc.txt
import csv, glob, collections, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("gpurun_out/%s_*" % tag)):
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "wire_ingest" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
for d in gpurun_out/${TAG}_*/; do rm -rf "$d"; done
