#!/bin/bash
# r2h: the bench lines of the final build (192 x 6 side blocks) + the MultiPaxos device tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mp_gpu.py tests/test_baseline_configs_gpu.py tests/test_spread_mp.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee gpurun_out/r2h_tests.log
timeout 900 python bench.py > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_bench.err
timeout 300 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra > gpurun_out/r2h_bench_steady.json 2>> gpurun_out/r2h_bench.err
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra > gpurun_out/r2h_bench_driver_flags_$i.json 2>> gpurun_out/r2h_bench.err; done
timeout 300 python bench.py --batch 0 --straggler-ticks 8 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra > gpurun_out/r2h_bench_per_tick_call_driver_flags.json 2>> gpurun_out/r2h_bench.err
for f in r2h_bench r2h_bench_steady r2h_bench_driver_flags_1 r2h_bench_driver_flags_2 r2h_bench_per_tick_call_driver_flags; do python - gpurun_out/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1].split("/")[-1], "value %.3e ms %.4f"%(d["value"],d["ms_per_step"]), "tally frac %.3f us %.1f"%(r["frac"], r["avg_launch_us"]), "whole_tick frac_pmc", round(r["whole_tick"]["frac_pmc"] or 0,3), {n: round(v.get("avg_us") or 0,1) for n,v in d["kernels"].items()})
PY
done
