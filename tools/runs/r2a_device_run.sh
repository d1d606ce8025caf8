#!/bin/bash
# Round 2, first device call: the whole -m gpu suite WITHOUT -x (every failure listed, not just the first), the legs of
# the kernels that had no device run in round 1, then the prepared compile-time A/Bs (tools/ab_experiments.sh).
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r2a_gputests.log
tail -3 gpurun_out/r2a_gputests.log
for leg in epaxos_execution rspaxos_replica craft_leader quorum_read; do
    timeout 200 python bench.py --leg $leg > gpurun_out/r2a_leg_$leg.json 2> gpurun_out/r2a_leg_$leg.err || echo "leg $leg failed"
    tail -c 400 gpurun_out/r2a_leg_$leg.json
done
timeout 1200 bash tools/ab_experiments.sh 2>&1 | tee gpurun_out/r2a_ab.log
