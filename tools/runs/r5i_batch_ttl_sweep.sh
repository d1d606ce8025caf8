#!/bin/bash
# the batch length / straggler ttl of the headline run, swept again on round 4's kernels (the bulk tick went from 0.071 to 0.056 ms
# steady: is the side launch of a batch now the longer of the two?)
B="--no-l2 --no-extra --no-rs --no-cpu --gpus 1 --steps 20 --warmup 5"
for rep in 1 2; do
for batch in 8 12 16; do
  for ttl in 2 4 8; do
    timeout 200 python bench.py $B --batch $batch --straggler-ticks $ttl > gpurun_out/r5i_b${batch}_t${ttl}_$rep.json 2>> gpurun_out/r5i.err
  done
done
done
python - <<P
import json
for batch in (8, 12, 16):
    for ttl in (2, 4, 8):
        v = []
        for rep in (1, 2):
            try:
                d = json.loads(open("gpurun_out/r5i_b%d_t%d_%d.json" % (batch, ttl, rep)).read().strip().splitlines()[-1])
                v.append(d["ms_per_step"])
            except Exception as e:
                v.append(None)
        print("batch", batch, "ttl", ttl, "ms/tick", v)
P
