# round 6: do the two slices' launches overlap?  per-launch timelines of tools/experiments/ep_slices_probe.py 2 (kernel trace): the shipped library, and the
# one-by-one launch built without LDS (-DEPC_CL_NOLDS)
mkdir -p gpurun_out; R=$PWD
for tag in shipped clnolds; do
  if [ $tag = shipped ]; then unset SUMMERSET_HIP_LIB; else export SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_$tag.so; fi
  ( cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R; timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/s36_prof_$tag -- python $R/tools/experiments/ep_slices_probe.py 2 > /dev/null 2>&1 )
  python tools/rocpd_timeline.py gpurun_out/s36_prof_$tag ep_cluster --limit 4000 > gpurun_out/s36_timeline_$tag.txt 2>&1
  rm -rf gpurun_out/s36_prof_$tag
  echo "== $tag"; sed -n 20,44p gpurun_out/s36_timeline_$tag.txt
done
