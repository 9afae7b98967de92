# development: parity of the tail form, then kernel stats + timings of the online_tail schedule
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tailprof
timeout 600 python -m pytest tests -m gpu -x -q -k "tail" 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tailprof -- python3 tools/bench_learning_loop.py --steps 80 --warmup 40 --schedule online_tail > gpurun_out/tailprof/out.json 2>gpurun_out/tailprof/err.log
python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/tailprof/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
find gpurun_out/tailprof -name "*.db" -delete; find gpurun_out/tailprof -name "*kernel_trace.csv" -delete
for sch in online_tail online; do
timeout 200 python tools/bench_learning_loop.py --steps 200 --warmup 40 --schedule $sch 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['shares']
print('$sch pass_ms %.4f solve %.4f refit/step %.4f other %.4f  ms_per_step %.4f  value %.3f M/s  fails %s' % (s['pass_ms_per_step'], s['solve_ms_per_step'], s['refit_ms_per_step'], s['other_ms_per_step'], d['ms_per_step'], d['value'] / 1e6, d.get('append_or_refit_failures')))"
done
