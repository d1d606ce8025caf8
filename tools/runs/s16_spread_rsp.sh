mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_spread_rsp.py -m gpu -x -q > gpurun_out/s16_spread_rsp_tests.log 2>&1; tail -3 gpurun_out/s16_spread_rsp_tests.log
timeout 600 python bench.py --layout spread-rspaxos --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('library tick:', d['ms_per_step'], d['value'])"
SMR_L2_PYTHON_TICK=1 timeout 600 python bench.py --layout spread-rspaxos --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('python tick:', d['ms_per_step'], d['value'])"
