# r7m: last check of the round's final tree -- the payload store's tests, the C++ examples, the ABI tests, the driver's exact command, smoke()
mkdir -p gpurun_out
T=r7m
timeout 300 python -m pytest tests/test_zz_rsp_payload_gpu.py tests/test_zzz_example_rsp_payload_gpu.py tests/test_example_gpu.py tests/test_rs_gpu.py tests/test_zz_rsp_steady_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/${T}_tests.log; tail -2 gpurun_out/${T}_tests.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_command.json 2> gpurun_out/${T}_bench.err; python - <<P
import json
d = json.loads(open("gpurun_out/${T}_bench_driver_command.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.4f tally frac %.3f legs_failed %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("legs_failed")))
x = d.get("rspaxos_payload", {})
print("rspaxos_payload", x.get("ms_per_tick"), x.get("bytes_path_ms_per_tick"), x.get("verified"), x.get("error"), (x.get("roofline") or {}).get("frac"))
P
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
