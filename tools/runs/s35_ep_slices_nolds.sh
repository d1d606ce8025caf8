# round 6: the EPaxos groups as 1 / 2 / 4 clusters on streams of their own (tools/experiments/ep_slices_probe.py) with the one-by-one launch WITHOUT LDS
# (-DEPC_CL_NOLDS: its blocks fit beside a block of the batched kernel), against the shipped library; the config-5 device tests on the variant first
mkdir -p gpurun_out
export PYTHONPATH=$PWD
V=$PWD/summerset_amd/variants/libsummerset_hip_clnolds.so
SUMMERSET_HIP_LIB=$V timeout 600 python -m pytest tests/test_zzz_ep_cluster_fused_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu -k "ep or config4 or config5" -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do
  echo "shipped:"; timeout 300 python tools/experiments/ep_slices_probe.py 1 2 4 2>&1 | grep slices | tail -3
  echo "nolds:"; SUMMERSET_HIP_LIB=$V timeout 300 python tools/experiments/ep_slices_probe.py 1 2 4 2>&1 | grep slices | tail -3
done
