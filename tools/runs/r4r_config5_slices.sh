#!/bin/bash
# r4r: config 5 through the one-launch cluster tick at 65 536 groups against oracle slices, both orders
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -q -p no:cacheprovider -k "config5 or config4" --durations=4 2>&1 | tail -8 | tee gpurun_out/r4r.log
