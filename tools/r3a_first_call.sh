#!/bin/bash
# Round 3, first device call: what round 2 wrote after its GPU minutes were spent and the full record of the build
# it left -- the whole -m gpu suite WITHOUT -x (the six last-sorted files have never run on a device; one failing must not hide
# the others), the bench line (its epaxos_cluster.one_call_per_tick is smr_ep_cluster_tick's first number), the driver's flags,
# the two spread layouts with virtual ranks, the kernel trace of the default bench line.  ~6 GPU-minutes.
TAG=${1:-r3a}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rxX -p no:cacheprovider --durations=8 2>&1 | tail -80 > gpurun_out/${TAG}_gputests.log
tail -3 gpurun_out/${TAG}_gputests.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 600 gpurun_out/${TAG}_bench.json; echo
TAG=$TAG python - <<'P'
import json, os
try:
    d = json.loads(open("gpurun_out/%s_bench.json" % os.environ["TAG"]).read().strip().splitlines()[-1])
    print("value", d["value"], "ms/tick", d["ms_per_step"], "tally frac", d["roofline"]["frac"], "legs_failed", d.get("legs_failed"))
    print("epaxos_cluster", {k: d["epaxos_cluster"].get(k) for k in ("value", "ms_per_tick", "one_call_per_tick")})
    print("wire_ingest", d.get("wire_ingest", {}).get("call_us"), d.get("wire_ingest", {}).get("roofline", {}).get("frac"))
except Exception as e:
    print("bench line unreadable:", e)
P
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra > gpurun_out/${TAG}_bench_driver_flags.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra > gpurun_out/${TAG}_bench_steady.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --layout spread --spread-ranks 4 --steps 24 --warmup 6 > gpurun_out/${TAG}_bench_spread4.json 2> gpurun_out/${TAG}_bench_spread.err
timeout 300 python bench.py --layout spread-epaxos --spread-ranks 4 --steps 8 --warmup 2 > gpurun_out/${TAG}_bench_spread_epaxos4.json 2>> gpurun_out/${TAG}_bench_spread.err
timeout 300 python bench.py --layout colocated-epaxos --steps 20 --warmup 4 > gpurun_out/${TAG}_bench_colocated_epaxos.json 2>> gpurun_out/${TAG}_bench_spread.err; tail -c 700 gpurun_out/${TAG}_bench_colocated_epaxos.json; echo
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_bench -- python $R/bench.py --no-cpu --no-rs --no-extra > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /dev/null
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_epc -- python $R/bench.py --leg epaxos_cluster > $R/gpurun_out/${TAG}_leg_epaxos_cluster.json 2> /dev/null
cd $R
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_prof_bench -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats_default_bench.txt 2>&1
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_prof_epc -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats_epaxos_cluster.txt 2>&1
rm -rf gpurun_out/${TAG}_prof_bench gpurun_out/${TAG}_prof_epc
grep -v "at::native" gpurun_out/${TAG}_kernel_stats_default_bench.txt | head -12 | cut -c1-150
grep -v "at::native" gpurun_out/${TAG}_kernel_stats_epaxos_cluster.txt | head -14 | cut -c1-150
