#!/bin/bash
# what bounds ep_cluster_tick_kernel: SQ counters over the EPaxos cluster leg (kernel-trace + pmc only, one small set per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/r5s_$tag -- python $R/bench.py --leg epaxos_cluster > /dev/null 2>$R/gpurun_out/r5s_$tag.err
  python - <<P
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$R/gpurun_out/r5s_$tag/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "ep_cluster_tick" not in k: continue
        a = acc[(k[:60], row["Grid_Size"])][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
for k, d in acc.items():
    print(k, {c: (n, round(s / n)) for c, (n, s) in d.items()})
P
  rm -rf $R/gpurun_out/r5s_$tag
done
