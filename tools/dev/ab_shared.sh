cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "shared or regime or rollout or c4 or matern" 2>&1 | tail -2
for r in 1 2; do for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_prev.so; do
  echo "== $lib"
  for dt in f32 f64; do
  BCBF_LIB_PATH=$PWD/$lib python bench.py --regime shared --dtype $dt --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('shared $dt %.3f M  ms %.4f frac %.4f kernel_ms %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms']))"
  BCBF_LIB_PATH=$PWD/$lib python bench.py --regime shared --dtype $dt --parts 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('shared $dt parts1 %.3f M  ms %.4f frac %.4f kernel_ms %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms']))"
  done
done; done
