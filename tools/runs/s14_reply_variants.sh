mkdir -p gpurun_out
for v in "" rw512 rw256; do
  if [ -n "$v" ]; then export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$v.so; fi
  timeout 400 python bench.py --leg reply_ingest 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['frames_to_last_commit'])"
done
