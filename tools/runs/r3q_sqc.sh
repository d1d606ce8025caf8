#!/bin/bash
# r3q: instruction-cache / scalar-cache counters of the one-launch EPaxos tick
mkdir -p gpurun_out
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_LEVEL[A-Z_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/r3q_counter_names.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/r3q_$tag -- python $R/bench.py --leg epaxos_cluster > /dev/null 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/r3q_sqc.txt
import csv, glob, collections
print(open("gpurun_out/r3q_counter_names.txt").read()[:1500])
for d in sorted(glob.glob("gpurun_out/r3q_S*")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "ep_cluster_tick" in r["Kernel_Name"] or "ep_acceptor_kernel<0>" in r["Kernel_Name"] or "ep_execute" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k[:44].ljust(44), "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
rm -rf gpurun_out/r3q_S*
