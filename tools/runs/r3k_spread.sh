#!/bin/bash
# r3k: the spread (L2) tick through smr_mp_spread_segment against the call-by-call tick, 4 and 8 virtual ranks; the l2 object in the driver's line
mkdir -p gpurun_out
{ timeout 600 python -m pytest tests/test_spread_mp.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for n in 4 8; do
SMR_SPREAD_AB=1 timeout 300 python bench.py --layout spread --spread-ranks $n --steps 24 --warmup 6 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r3k_bench_spread$n.json
python - <<P
import json
d=json.loads(open("gpurun_out/r3k_bench_spread$n.json").read())
print("ranks $n: ms/tick", round(d["ms_per_step"],4), "call by call", round(d["call_by_call"]["ms_per_step"],4), "bytes per exchange", d["exchange"]["bytes_per_exchange_per_rank"])
P
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra 2>gpurun_out/r3k_bench.err | tail -1 > gpurun_out/r3k_bench_driver_flags.json
python - <<P
import json
d=json.loads(open("gpurun_out/r3k_bench_driver_flags.json").read())
print("driver flags: value", d["value"], "ms/step", d["ms_per_step"], "regions", [round(x,4) for x in d["timed_regions"]["ms_per_step"]])
print("l2:", json.dumps(d.get("l2"))[:900])
P
tail -3 gpurun_out/r3k_bench.err
} 2>&1 | tee gpurun_out/r3k.log
