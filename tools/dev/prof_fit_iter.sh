# Development: which kernels an Adam iteration of the batched fit spends its time in (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/prof_fit_iter
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/dev/try_fit_scale.py > $O/out.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_kernel_trace.csv" -size +2M -delete 2>/dev/null
f=$(find $O -name "*kernel_stats.csv" | head -1)
head -14 $f | cut -c1-220
