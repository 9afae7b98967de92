# A/B (development): twice the columns per stage in the B-side-only blocks (BCBF_PS_DBL_B, fp64) against the plain B-only loop
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or c3 or c5 or tail or reserved or jets or reldeg or potrs or query or c2" 2>&1 | tail -3
for round in 1 2; do
  for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_nodbl.so; do
    echo "== $lib"
    BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_online.py --n0 1024 --n1 2048 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['segments'][0]
print('C5 b256 append_ms %.4f frac %.3f' % (s['append_ms'], s['roofline']['frac']))"
    BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_online.py --batch 1024 --n0 1024 --n1 1280 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['segments'][0]
print('C5 b1024 append_ms %.4f frac %.3f' % (s['append_ms'], s['roofline']['frac']))"
    BCBF_LIB_PATH=$PWD/$lib timeout 300 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    print(d.get('config'), {k: round(v, 4) for k, v in d.items() if k.endswith('_ms')}, {k: round(v.get('frac', 0) or 0, 3) for k, v in d.get('roofline', {}).items() if isinstance(v, dict)})"
    BCBF_LIB_PATH=$PWD/$lib python bench.py --cpu-sample 0 --dtype f64 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('headline f64 %.3f M  ms %.4f frac %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac']))"
  done
done
