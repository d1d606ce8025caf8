mkdir -p gpurun_out
R=$PWD
( cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s5_prof -- python $R/bench.py --leg epaxos_cluster > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/s5_prof > gpurun_out/s5_kernel_stats.txt 2>&1
grep -i "ep_cluster\|commit_one" gpurun_out/s5_kernel_stats.txt | cut -c1-220
rm -rf gpurun_out/s5_prof
