# A/B of libbcbf variants on the online learning loop's fused pass (development)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_*.so; do
    BCBF_LIB_PATH=$PWD/$lib timeout 120 python tools/bench_learning_loop.py --steps 80 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', 'pass_ms %.4f  ms_per_step %.4f' % (d['shares']['pass_ms_per_step'], d['ms_per_step']))
"
  done
done
