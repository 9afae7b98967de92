# development: the row-major tail form of the online schedule -- parity tests, then the three schedules of the learning loop
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "tail or learning or window or reserved" 2>&1 | tail -15 > gpurun_out/ab_tail_tests.log
for round in 1 2; do
  for sch in online online_tail reference; do
    timeout 200 python tools/bench_learning_loop.py --steps 200 --warmup 40 --schedule $sch 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['shares']
print('$sch pass_ms %.4f solve %.4f refit/step %.4f other %.4f  ms_per_step %.4f  value %.3f M/s  fails %s' % (s['pass_ms_per_step'], s['solve_ms_per_step'], s['refit_ms_per_step'], s['other_ms_per_step'], d['ms_per_step'], d['value'] / 1e6, d.get('append_or_refit_failures')))" >> gpurun_out/ab_tail.log
  done
done
