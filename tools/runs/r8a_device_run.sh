#!/bin/bash
# round 5, first call: the driver's exact command on the round's first tree -- is the final stdout line small and parseable?
mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r8a_bench_stdout.txt 2> gpurun_out/r8a_bench.err
echo "rc $?"; tail -n 1 gpurun_out/r8a_bench_stdout.txt | wc -c; tail -n 1 gpurun_out/r8a_bench_stdout.txt
cp bench_detail.json gpurun_out/r8a_bench_detail.json 2>/dev/null
tail -3 gpurun_out/r8a_bench.err
