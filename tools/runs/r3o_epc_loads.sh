#!/bin/bash
# r3o: memory instructions per wavefront of the one-launch EPaxos tick, execution on / off
mkdir -p gpurun_out
R=$PWD; cd /tmp; export TMPDIR=/tmp
for ex in 1 0; do
SMR_EPC_EXECUTE=$ex timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r3o_$ex -- python $R/bench.py --leg epaxos_cluster > /dev/null 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/r3o_epc_loads.txt
import csv, glob, collections
for d in ("r3o_1", "r3o_0"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob("gpurun_out/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "ep_cluster_tick" in r["Kernel_Name"] or "ep_execute" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print("execute=%s" % d[-1], k[:44].ljust(44), "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
rm -rf gpurun_out/r3o_1 gpurun_out/r3o_0
