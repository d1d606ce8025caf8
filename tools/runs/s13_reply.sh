mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_zz_reply_ingest_gpu.py tests/test_raft_gpu.py -m gpu -x -q > gpurun_out/s13_reply_tests.log 2>&1; tail -3 gpurun_out/s13_reply_tests.log
timeout 400 python bench.py --leg reply_ingest > gpurun_out/s13_leg_reply_ingest.json 2> gpurun_out/s13_leg_reply_ingest.err; tail -2 gpurun_out/s13_leg_reply_ingest.err
python - <<P
import json
d=json.loads(open("gpurun_out/s13_leg_reply_ingest.json").read().strip().splitlines()[-1])
print({k:v for k,v in d.items() if k in ("call_us","frames_to_last_commit","emit_then_ingest_us")}); print(d["roofline"]["frac"], d["ingest_alone"])
P
