#!/bin/bash
# SQ counters of the headline's kernels (tools/pmc_probe.py without --extra: the bench launch shape), one small set per pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/r6i_$tag -- python $R/tools/pmc_probe.py > /dev/null 2>$R/gpurun_out/r6i_$tag.err
  python - <<P
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$R/gpurun_out/r6i_$tag/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "mp_" not in k: continue
        a = acc[(k.split("(")[0][:40], row["Grid_Size"])][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
for k, d in sorted(acc.items()):
    print(k, {c: (n, round(s / n)) for c, (n, s) in d.items()})
P
  rm -rf $R/gpurun_out/r6i_$tag
done
