#!/bin/bash
mkdir -p gpurun_out
for lib in "" $PWD/summerset_amd/variants/libsummerset_hip_raftl1.so; do
  if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
  timeout 300 python -m pytest tests/test_raft_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -1
  for i in 1 2; do
  timeout 300 python bench.py --leg raft > gpurun_out/r2s.json 2> gpurun_out/r2s.err || timeout 300 python - > gpurun_out/r2s.json 2> gpurun_out/r2s.err <<'PY'
import json, torch, bench
print(json.dumps(bench.raft_leg(torch, torch.device("cuda"))))
PY
  python - "lib=$(basename "$lib")" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r2s.json").read().strip().splitlines()[-1])
print(sys.argv[1], "raft replies us %.2f frac %.3f tick us %.1f" % (d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["us_per_tick"]))
PY
  done
done 2>&1 | tee gpurun_out/r2s_raft_ab.log
