#!/bin/bash
# r4p: the emit kernels' tests on the device (+ reply ingest incl. RSPaxos, the 5-exchange EPaxos L2 schedule with execution, role rotation);
# the RS leg with the sweep's from_data + encode in one pass
mkdir -p gpurun_out
{ timeout 900 python -m pytest tests/test_zz_wire_emit_gpu.py tests/test_zz_reply_ingest_gpu.py tests/test_zzy_spread_ep_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout 600 python bench.py --no-cpu --no-extra --no-l2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['rs_encode']
print('rs_encode', round(r['value'],1), r['unit'], 'frac', round(r['roofline']['frac'],3))
for x in r['rse_bench_sweep']: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in x.items()})
json.dump(r, open('gpurun_out/r4p_leg_rs_encode.json','w'))"
} 2>&1 | tee gpurun_out/r4p.log
