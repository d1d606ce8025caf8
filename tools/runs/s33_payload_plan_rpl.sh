# round 6: the payload stores' plan kernels with 8 / 16 ring rows per lane (-DPS_RPL_MAX) against the shipped 4: payload device tests per variant, the two legs x3, same box
mkdir -p gpurun_out
for tag in rpl8 rpl16; do
  SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$tag.so timeout 600 python -m pytest tests/test_zz_rsp_payload_gpu.py tests/test_zz_craft_payload_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2 > gpurun_out/s33_tests_$tag.log
  echo "$tag tests: $(tail -1 gpurun_out/s33_tests_$tag.log)"
done
for i in 1 2 3; do
  for tag in shipped rpl8 rpl16; do
    if [ $tag = shipped ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_$tag.so; fi
    for leg in rspaxos_payload craft_payload; do
      timeout 300 python bench.py --leg $leg > gpurun_out/s33_${leg}_${tag}_$i.json 2> gpurun_out/s33_${leg}_${tag}_$i.err
      python - $tag $leg gpurun_out/s33_${leg}_${tag}_$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    print(sys.argv[1], sys.argv[2], "ms/tick %.4f" % d["ms_per_tick"], "bytes path %.4f" % d.get("bytes_path_ms_per_tick", 0), "verified", d.get("verified"))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "failed:", e)
PY
    done
  done
done
unset SUMMERSET_HIP_LIB
