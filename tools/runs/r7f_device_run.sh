# r7f: the round's last device run -- the whole -m gpu suite and the driver's exact command on the final tree
mkdir -p gpurun_out
T=r7f
timeout 500 python -m pytest tests -m gpu -q -rxX -p no:cacheprovider --durations=6 2>&1 | tail -25 > gpurun_out/${T}_gputests.log; tail -3 gpurun_out/${T}_gputests.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_command.json 2> gpurun_out/${T}_bench.err; python - <<P
import json
d = json.loads(open("gpurun_out/${T}_bench_driver_command.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.4f tally frac %.3f legs_failed %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("legs_failed")))
x = d.get("rspaxos_payload", {})
print("rspaxos_payload", x.get("ms_per_tick"), x.get("bytes_path_ms_per_tick"), x.get("verified"), x.get("error"))
P
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
