#!/bin/bash
mkdir -p gpurun_out
{ timeout 200 python tools/exp_halves.py --parts 1 2 4 --timeouts 0
  timeout 200 python tools/exp_halves.py --parts 2 4 --timeouts 0 --threads
  timeout 200 python tools/exp_halves.py --parts 2 --timeouts 0 --batch 16
  timeout 200 python tools/exp_halves.py --parts 2 --timeouts 0 --batch 16 --threads; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2n_halves_steady.log
