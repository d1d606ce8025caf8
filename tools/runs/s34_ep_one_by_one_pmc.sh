# round 6: what the EPaxos one-by-one launch spends its time on -- SQ counters per kernel over tools/experiments/ep_slices_probe.py 1 (config 5, one cluster, 2 x 14 ticks)
mkdir -p gpurun_out; R=$PWD
( cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/s34_a -- python $R/tools/experiments/ep_slices_probe.py 1 > /dev/null 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/s34_b -- python $R/tools/experiments/ep_slices_probe.py 1 > /dev/null 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/s34_c -- python $R/tools/experiments/ep_slices_probe.py 1 > /dev/null 2>&1 )
python - <<'PY'
import csv, glob, collections
for tag in "abc":
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("gpurun_out/s34_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            if "ep_cluster" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(tag, k, {c: "%.4g per launch (%d launches)" % (v / n[(k, c)], n[(k, c)]) for c, v in d.items()})
PY
rm -rf gpurun_out/s34_a gpurun_out/s34_b gpurun_out/s34_c
