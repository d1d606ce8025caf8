# r7h: the payload leg's launch shape held against the oracles at its size; the leg again (now through workloads.py)
mkdir -p gpurun_out
T=r7h
timeout 300 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -x -q -p no:cacheprovider -k "payload_store" --durations=3 2>&1 | tail -8 > gpurun_out/${T}_payload_at_size.log; tail -4 gpurun_out/${T}_payload_at_size.log
timeout 200 python bench.py --leg rspaxos_payload > gpurun_out/${T}_leg_rspaxos_payload.json 2> gpurun_out/${T}_leg.err; tail -c 400 gpurun_out/${T}_leg_rspaxos_payload.json; tail -3 gpurun_out/${T}_leg.err
