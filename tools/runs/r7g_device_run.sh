# r7g: HBM bytes of the payload store's kernels -- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) over the leg
mkdir -p gpurun_out
T=r7g
R=$PWD
( cd /tmp && export TMPDIR=/tmp
  timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_pmc_fetch -- python $R/bench.py --leg rspaxos_payload > /dev/null 2>&1
  timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_pmc_write -- python $R/bench.py --leg rspaxos_payload > /dev/null 2>&1 )
python tools/pmc_traffic.py gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write "bench.py --leg rspaxos_payload at HEAD (config 4's shape: 16384 groups x L = 4113, window 16): per tick one ps_put_kernel<3>, five ps_plan_kernel + ps_bytes_kernel (the leader's and four followers'), one rsp_cluster_tick_kernel" > gpurun_out/${T}_pmc_traffic_payload_leg.json 2> gpurun_out/${T}_pmc.err
rm -rf gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write
python - <<P
import json
d = json.load(open("gpurun_out/${T}_pmc_traffic_payload_leg.json"))
for k, v in d["kernels"].items():
    print(k[:50], v["launches"], "read %.1f MB write %.1f MB" % (v["hbm_read_bytes_per_launch"] / 1e6, v["hbm_write_bytes_per_launch"] / 1e6))
P
tail -2 gpurun_out/${T}_pmc.err
