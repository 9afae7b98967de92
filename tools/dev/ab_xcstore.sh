cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "tail or learning or window or reserved or c5 or online or append" 2>&1 | tail -4
for sch in online_tail online; do
timeout 200 python tools/bench_learning_loop.py --steps 200 --warmup 40 --schedule $sch 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['shares']
print('$sch pass_ms %.4f solve %.4f refit/step %.4f other %.4f  ms_per_step %.4f  value %.3f M/s  fails %s' % (s['pass_ms_per_step'], s['solve_ms_per_step'], s['refit_ms_per_step'], s['other_ms_per_step'], d['ms_per_step'], d['value'] / 1e6, d.get('append_or_refit_failures')))"
done
for cfg in "256 1024 2048" "1024 1024 1280"; do
  set -- $cfg
  timeout 200 python tools/bench_online.py --batch $1 --n0 $2 --n1 $3 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['segments'][0]
print('C5 b$1 append_ms %.4f step_ms %.4f frac %.3f' % (s['append_ms'], s['step_ms'], s['roofline']['frac']))"
done
timeout 300 python tools/bench_online.py 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('C5 full', [(s['N_from'], s['N_to'], round(s['append_ms'], 4), round(s['step_ms'], 4)) for s in d['segments']])"
