#!/bin/bash
# Round 2, final build: the whole -m gpu suite, the bench line (default: batches of 8 ticks, straggler list on; and one call per tick beside it), the spread layout on one GPU
# (virtual ranks), then the rocprofv3 passes (kernel trace, FETCH_SIZE, WRITE_SIZE -- separate runs) over tools/pmc_probe.py
TAG=${1:-r2t}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -40 > gpurun_out/${TAG}_gputests.log
tail -3 gpurun_out/${TAG}_gputests.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 600 gpurun_out/${TAG}_bench.json; echo
timeout 300 python bench.py --timeouts 0 --no-cpu --no-rs --no-extra > gpurun_out/${TAG}_bench_steady.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra > gpurun_out/${TAG}_bench_driver_flags.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --batch 0 --straggler-ticks 8 --no-cpu --no-rs --no-extra > gpurun_out/${TAG}_bench_per_tick_call.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --batch 0 --straggler-ticks 8 --steps 20 --warmup 5 --no-cpu --no-rs --no-extra > gpurun_out/${TAG}_bench_per_tick_call_driver_flags.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --layout spread --spread-ranks 4 --steps 24 --warmup 6 > gpurun_out/${TAG}_bench_spread4.json 2> gpurun_out/${TAG}_bench_spread.err; tail -c 900 gpurun_out/${TAG}_bench_spread4.json; echo
timeout 300 python bench.py --layout spread --spread-ranks 8 --steps 24 --warmup 6 > gpurun_out/${TAG}_bench_spread8.json 2>> gpurun_out/${TAG}_bench_spread.err; tail -c 400 gpurun_out/${TAG}_bench_spread8.json; echo
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -- python $R/tools/pmc_probe.py --extra > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_bench -- python $R/bench.py --no-cpu --no-rs --no-extra > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python $R/tools/pmc_probe.py --extra > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python $R/tools/pmc_probe.py --extra > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_prof -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats.txt 2>&1
python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write "tools/pmc_probe.py --extra: 16 ticks of the bench shape on the default workload (65536 groups x 5, S=32, H=4, 10% loss, 1% leader changes) as bench.py runs them: two smr_mp_run_ticks batches of 8 with the straggler list on (ttl 4), 32 more through the fused tick kernel (2 launches of 16) + 3 RS(3,2) encodes of 65536 x 4099 B + the Raft / EPaxos legs" > gpurun_out/${TAG}_pmc_traffic.json 2> gpurun_out/${TAG}_pmc_traffic.err
python tools/rocpd_summary.py $(find gpurun_out/${TAG}_prof_bench -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats_default_bench.txt 2>&1
grep -v "at::native" gpurun_out/${TAG}_kernel_stats.txt | head -24 | cut -c1-150
grep -v "at::native" gpurun_out/${TAG}_kernel_stats_default_bench.txt | head -12 | cut -c1-150
