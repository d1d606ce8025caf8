mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s0_smoke.log 2>&1 || { echo SMOKE FAILED; tail -5 gpurun_out/s0_smoke.log; exit 3; }
SUMMERSET_HIP_LIB=summerset_amd/variants/libsummerset_hip_epc_stamps.so timeout 300 python tools/dbg_epc_stamps.py 1 pm > gpurun_out/s0_epc_stamps_pm.log 2>&1
timeout 400 python bench.py --leg epaxos_cluster > gpurun_out/s0_leg_epaxos_cluster.json 2> gpurun_out/s0_leg_epaxos_cluster.err
tail -c 1500 gpurun_out/s0_leg_epaxos_cluster.json
grep -v "^  q" gpurun_out/s0_epc_stamps_pm.log | head -30
