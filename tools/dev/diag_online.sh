# development: which part of the online learning loop makes its fused pass slower than the same call in isolation
cd $GRAFT_REPO_ROOT
for e in none noevents none noevents; do
  BCBF_LEARN_EXPERIMENT=$e timeout 120 python tools/bench_learning_loop.py --steps 200 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['shares']
print('%-28s ms_per_step %.4f  value %.3f M/s' % ('$e', d['ms_per_step'], d['value'] / 1e6))
"
done
