# round 6: per-kernel times with / without the fold, steady state and the driver's command (rocprofv3 --kernel-trace --stats)
# (when this ran the fold was on by default and SMR_MP_NO_FOLD_R1 turned it off; since then it is opt-in: SMR_MP_FOLD_R1)
mkdir -p gpurun_out
for mode in steady driver; do
  for v in nofold fold fold_defer; do
    unset SMR_MP_NO_FOLD_R1 SMR_MP_ALWAYS_DEFER_REST
    [ $v = nofold ] && export SMR_MP_NO_FOLD_R1=1
    [ $v = fold_defer ] && export SMR_MP_ALWAYS_DEFER_REST=1
    if [ $mode = steady ]; then args="--timeouts 0"; else args="--gpus 1 --steps 20 --warmup 5"; fi
    ( cd /tmp && export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s19_prof -- python $GRAFT_REPO_ROOT/bench.py $args --no-cpu --no-rs --no-extra --no-l2 > $GRAFT_REPO_ROOT/gpurun_out/s19_${v}_${mode}.json 2>/dev/null )
    python tools/rocpd_summary.py gpurun_out/s19_prof > gpurun_out/s19_kernel_stats_${v}_${mode}.txt 2>&1; rm -rf gpurun_out/s19_prof
    echo "== $v $mode"; grep "smr::" gpurun_out/s19_kernel_stats_${v}_${mode}.txt | head -8 | cut -c1-40,75-140
  done
done
