cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or c3 or c1 or c2 or facade or bench" 2>&1 | tail -2
for r in 1 2; do for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_prev.so; do
  echo "== $lib"
  BCBF_LIB_PATH=$PWD/$lib python bench.py --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('headline %.3f M  ms %.4f frac %.4f kernel_ms %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms']))"
  BCBF_LIB_PATH=$PWD/$lib python bench.py --cpu-sample 0 --parts 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('headline parts1 %.3f M  ms %.4f frac %.4f kernel_ms %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms']))"
  BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_learning_loop.py --steps 120 --warmup 40 --schedule reference --parts 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('learn reference parts4 pass_ms %.4f value %.3f M/s' % (d['shares']['pass_ms_per_step'], d['value'] / 1e6))"
done; done
