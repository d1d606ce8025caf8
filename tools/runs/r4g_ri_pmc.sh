#!/bin/bash
# r4g: counters of the reply ingest kernel
mkdir -p gpurun_out
R=$PWD; export PYTHONPATH=$R
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_GDS SQ_INSTS_FLAT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/r4g_$tag -- python $R/bench.py --leg reply_ingest > /tmp/pmc_$tag.log 2>&1 || tail -3 /tmp/pmc_$tag.log
done
cd $R
python - <<'PY' | tee gpurun_out/r4g_ri_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("gpurun_out/r4g_*")):
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "wire_ingest" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
for d in gpurun_out/r4g_*/; do rm -rf "$d"; done
