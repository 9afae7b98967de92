cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tailprof
for v in 0 1; do
  if [ $v = 1 ]; then export BCBF_TAIL_NOWFULL=1; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tailprof/v$v -- python3 tools/dev/micro_tail.py > gpurun_out/tailprof/out$v.txt 2>gpurun_out/tailprof/err$v.log
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/tailprof/v$v/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:3]:
    print("nowfull=$v", r["Name"][:80], r["Calls"], r["AverageNs"])
PY
done
find gpurun_out/tailprof -name "*.db" -delete; find gpurun_out/tailprof -name "*kernel_trace.csv" -delete
