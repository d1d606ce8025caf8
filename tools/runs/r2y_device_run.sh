#!/bin/bash
# r2y: first device run of the lease manager and EPaxos explicit prepare kernels, plus the whole device suite
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5 | tee gpurun_out/r2y_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2y_bench_driver_flags.json 2> gpurun_out/r2y_bench.err
tail -c 600 gpurun_out/r2y_bench_driver_flags.json
