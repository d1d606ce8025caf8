#!/bin/bash
# r4n: role rotation of the bulk round launches (smr_mp_set_role_rotation): its parity test on the device, then the headline at the
# driver's flags and at the default flags, rotation off / on, twice each
mkdir -p gpurun_out
{ timeout 900 python -m pytest tests/test_mp_gpu.py -m gpu -q -p no:cacheprovider -k "role_rotation or batched or bench_shape" 2>&1 | tail -3
for rep in 1 2; do for rot in 0 1; do
  for flags in "--gpus 1 --steps 20 --warmup 5" ""; do
    timeout 300 python bench.py $flags --no-cpu --no-rs --no-extra --no-l2 --role-rotation $rot 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('rot $rot flags[$flags]', 'ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], ' '.join('%s %.1f' % (a, b['avg_us']) for a, b in k.items()))"
  done
done; done
} 2>&1 | tee gpurun_out/r4n.log
