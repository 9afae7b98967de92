# development: which part of the online learning loop makes its fused pass slower than the same call in isolation
cd $GRAFT_REPO_ROOT
for e in none nosolve nofails nosolve,nofails plainquery fixedN fixedN,nosolve,nofails; do
  BCBF_LEARN_EXPERIMENT=$e timeout 120 python tools/bench_learning_loop.py --steps 80 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['shares']
print('%-28s pass %.4f solve %.4f other %.4f ms_per_step %.4f' % ('$e', s['pass_ms_per_step'], s['solve_ms_per_step'], s['other_ms_per_step'], d['ms_per_step']))
"
done
