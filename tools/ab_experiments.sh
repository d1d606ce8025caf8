#!/bin/bash
# Device A/B of the prepared kernel experiments (tools/experiments/README.md), one gpurun call:
#   python tools/build_variant.py regout -DSMR_SKIP_REG_OUTBOX          (here, before the call: the .so travels)
#   python tools/build_variant.py ackbits -DSMR_ACK_BITS
#   python tools/build_variant.py balrun -DSMR_BAL_RUN
#   python tools/build_variant.py ballazy -DSMR_BAL_RUN -DSMR_BAL_LAZY
#   python tools/build_variant.py statuslazy -DSMR_BAL_RUN -DSMR_STATUS_LAZY
#   python tools/build_variant.py all5 -DSMR_ACK_BITS -DSMR_SKIP_REG_OUTBOX -DSMR_BAL_RUN -DSMR_BAL_LAZY -DSMR_STATUS_LAZY
#   gpurun --timeout 1500 -- 'bash tools/ab_experiments.sh'
# Per variant: the MultiPaxos device tests (parity first), then the headline bench line twice; results in gpurun_out/ab_*.
mkdir -p gpurun_out
run() {   # tag, library path ("" = the shipped one)
    local tag=$1 lib=$2
    if [ -n "$lib" ]; then export SUMMERSET_HIP_LIB=$lib; else unset SUMMERSET_HIP_LIB; fi
    timeout 500 python -m pytest tests/test_mp_gpu.py -x -q -m gpu > gpurun_out/ab_${tag}_tests.log 2>&1
    echo "$tag tests: $(tail -1 gpurun_out/ab_${tag}_tests.log)"
    for i in 1 2; do
        timeout 300 python bench.py --no-cpu --no-rs --no-extra > gpurun_out/ab_${tag}_bench$i.json 2> gpurun_out/ab_${tag}_bench$i.err
        python - "$tag" gpurun_out/ab_${tag}_bench$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d.get("kernels", {})
    print(sys.argv[1], "value %.3e" % d["value"], "ms/tick %.4f" % d["ms_per_step"], "roofline frac %.3f" % d["roofline"]["frac"],
          {n: round(v.get("avg_us", 0), 1) for n, v in k.items()} if isinstance(k, dict) else "")
except Exception as e:
    print(sys.argv[1], "bench failed:", e)
PY
    done
}
run shipped ""
for tag in regout ackbits balrun ballazy statuslazy all5; do
    lib=$PWD/summerset_amd/variants/libsummerset_hip_$tag.so
    [ -f "$lib" ] && run $tag "$lib" || echo "$tag: build it first (tools/build_variant.py)"
done
