# HBM traffic and kernel trace of the online path (BASELINE configs[4]): the fused append + control-query pass
# (`posterior_step_kernel<double, 3, 4, 0, 1, false, 1>`) and the in-place row writer (`gp_append_inplace_kernel`) at
# N = 1024 -> 2048, batch 256, and N = 1024 -> 1280, batch 1024.  Three PMC passes (counters + kernel trace only, the
# program directly after `--`), one `--kernel-trace --stats` pass; summary -> gpurun_out/$R/pmc_traffic_append.json.
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
for cfg in "256 1024 2048" "1024 1024 1280"; do
  set -- $cfg
  tag=b$1_n$2_$3
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d=$O/pmc_append_${tag}_$(echo $c | tr ' ' '_')
    rm -rf $d
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/bench_online.py --batch $1 --n0 $2 --n1 $3 > $d.log 2>&1
  done
  d=$O/prof_append_${tag}
  rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_online.py --batch $1 --n0 $2 --n1 $3 > $O/online_${tag}_prof.json 2> $d.err
  python3 tools/bench_online.py --batch $1 --n0 $2 --n1 $3 > $O/online_${tag}.json 2>/dev/null
done
python3 tools/summarise_pmc_append.py $O > $O/pmc_traffic_append.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_kernel_trace.csv" -size +2M -delete 2>/dev/null
cat $O/pmc_traffic_append.txt | tail -5
