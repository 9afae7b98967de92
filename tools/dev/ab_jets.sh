# A/B of the jets kernel variants under tools/_variants (development): the rel-degree-2 bench per library + the headline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "jets or reldeg or cbc2 or rel_degree or parity" 2>&1 | tail -3
for round in 1 2; do
  for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_*.so; do
    echo "== $lib"
    BCBF_LIB_PATH=$PWD/$lib timeout 150 python tools/bench_reldeg2.py 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['N'], d['n'], d['m'], d['dtype'], 'jets_ms %.4f GB/s %.0f values_ms %.4f' % (d['jets_ms'], d['jets_GBs_algorithmic'], d['values_only_ms']))"
    BCBF_LIB_PATH=$PWD/$lib python bench.py --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('headline %.3f M  ms %.4f frac %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac']))"
  done
done
