#!/bin/bash
# EPaxos cluster tick: time and HBM traffic (FETCH_SIZE / WRITE_SIZE passes) with execution on and off
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ex in 1 0; do
  export SMR_EPC_EXECUTE=$ex
  timeout 300 python $R/bench.py --leg epaxos_cluster > $R/gpurun_out/r5v_leg_exec$ex.json 2>> $R/gpurun_out/r5v.err
  python - <<P
import json
d = json.loads(open("$R/gpurun_out/r5v_leg_exec$ex.json").read().strip().splitlines()[-1])
print("execute=$ex", {k: round(v["ms_per_tick"], 4) for k, v in d.items() if isinstance(v, dict) and "ms_per_tick" in v})
P
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r5v_$c -- python $R/bench.py --leg epaxos_cluster > /dev/null 2>>$R/gpurun_out/r5v.err
    python - <<P
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$R/gpurun_out/r5v_$c/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "ep_cluster_tick" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
for k, (n, s) in acc.items(): print("execute=$ex", k, n, "launches, avg", round(s / n / 1e3, 1), "MB-ish per launch (KB units: x2 for FETCH on gfx950 per the guide)")
P
    rm -rf $R/gpurun_out/r5v_$c
  done
done
