# round 6: register caps on the cooperative side kernel (mp_straggler_batch at 212 VGPRs): STRAG_BATCH_MINW 3 / 4 (168 / 128 VGPRs + spills) against the
# shipped one, the driver's command, same box.  (Round 2 measured this on the per-lane side kernel: the side launch became the bottleneck.)
mkdir -p gpurun_out
for i in 1 2 3; do
  for tag in shipped sb3 sb4; do
    if [ $tag = shipped ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$tag.so; fi
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra > gpurun_out/s32_${tag}_$i.json 2> gpurun_out/s32_${tag}_$i.err
    python - $tag gpurun_out/s32_${tag}_$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/tick %.4f" % d["ms_per_step"], "value %.3e" % d["value"])
except Exception as e:
    print(sys.argv[1], "bench failed:", e)
PY
  done
done
unset SUMMERSET_HIP_LIB
