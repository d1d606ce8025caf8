#!/bin/bash
# r3f: where the one-launch EPaxos cluster tick's wave cycles go: SQ counters (two passes), L2 hit/miss, HBM traffic (FETCH / WRITE passes)
mkdir -p gpurun_out
R=$PWD; cd /tmp; export TMPDIR=/tmp
P="python $R/bench.py --leg epaxos_cluster"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/r3f_sq1 -- $P > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/r3f_sq2 -- $P > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum --output-format csv -d $R/gpurun_out/r3f_tcc -- $P > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3f_fetch -- $P > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3f_write -- $P > /dev/null 2>&1
cd $R
python - <<'PY' | tee gpurun_out/r3f_epc_pmc.txt
import csv, glob, collections
for d in ("r3f_sq1", "r3f_sq2", "r3f_tcc", "r3f_fetch", "r3f_write"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob("gpurun_out/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "smr::ep_" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(d, k[:44].ljust(44), "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())), " n=%d" % max(len(v) for v in acc[k].values()))
PY
rm -rf gpurun_out/r3f_sq1 gpurun_out/r3f_sq2 gpurun_out/r3f_tcc gpurun_out/r3f_fetch gpurun_out/r3f_write
