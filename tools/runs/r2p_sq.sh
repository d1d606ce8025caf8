#!/bin/bash
# SQ counters (own pass, --kernel-trace only) of the latency-bound kernels: where do the wave cycles go
mkdir -p gpurun_out
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/r2p_sq -- python $R/tools/pmc_probe.py --extra --timeouts 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r2p_sq2 -- python $R/tools/pmc_probe.py --extra --timeouts 0 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r2p_sq", "gpurun_out/r2p_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "smr::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k[:48].ljust(48), "  ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
