cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "learning" 2>&1 | tail -4
for args in "--schedule online_tail" "--schedule online_tail --parts 2" "--schedule online_tail --parts 4" "--schedule online --parts 4" "--schedule reference --parts 4"; do
timeout 200 python tools/bench_learning_loop.py --steps 200 --warmup 40 $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['shares']
print('$args: pass_ms %.4f solve %.4f refit/step %.4f other %.4f  ms_per_step %.4f  value %.3f M/s  fails %s  final %s' % (s['pass_ms_per_step'], s['solve_ms_per_step'], s['refit_ms_per_step'], s['other_ms_per_step'], d['ms_per_step'], d['value'] / 1e6, d.get('append_or_refit_failures'), d['final_vs_fp64_refit_on_device']))"
done
