# A/B (development) of libbcbf variants on the online learning loop only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in 1 2; do
  for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_*.so; do
    BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_learning_loop.py --steps 200 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib learn online pass_ms %.4f  ms_per_step %.4f  value %.3f M/s' % (d['shares']['pass_ms_per_step'], d['ms_per_step'], d['value'] / 1e6))" >> gpurun_out/ab_learn.log
  done
done
