# A/B (development): B-side-only blocks for waves whose A-side rows are dead (BCBF_PS_SKIP_DEAD_A) against the plain loop
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or c3 or c5 or tail or reserved or jets or reldeg or potrs or query" 2>&1 | tail -4
for round in 1 2; do
  for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_noskip.so; do
    echo "== $lib"
    BCBF_LIB_PATH=$PWD/$lib python bench.py --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('headline %.3f M  ms %.4f frac %.4f kernel_ms %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms']))"
    BCBF_LIB_PATH=$PWD/$lib timeout 150 python tools/bench_reldeg2.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tail -3
    BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_learning_loop.py --steps 120 --warmup 40 --schedule online_tail 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('online_tail pass_ms %.4f value %.3f M/s' % (d['shares']['pass_ms_per_step'], d['value'] / 1e6))"
    BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_online.py --n0 1024 --n1 2048 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['segments'][0]
print('C5 b256 append_ms %.4f frac %.3f' % (s['append_ms'], s['roofline']['frac']))"
    BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_configs.py 2>/dev/null | cut -c1-400 | head -12
  done
done
