mkdir -p gpurun_out
R=$PWD
( cd /tmp && export TMPDIR=/tmp
  SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_$1.so timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s7_prof -- python $R/bench.py --leg epaxos_cluster > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/s7_prof > gpurun_out/s7_kernel_stats_$1.txt 2>&1
grep -i "ep_cluster\|commit_one" gpurun_out/s7_kernel_stats_$1.txt | cut -c1-220
rm -rf gpurun_out/s7_prof
