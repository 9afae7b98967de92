# A/B of libbcbf variants on the C5 growth bench (development): append + fused query, fp64
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "reserved or c5 or window or learning or online or tail" 2>&1 | tail -3
for round in 1 2; do
  for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_*.so; do
    echo "== $lib"
    for cfg in "256 1024 2048" "1024 1024 1280" "256 512 1024"; do
      set -- $cfg
      BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_online.py --batch $1 --n0 $2 --n1 $3 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['segments'][0]
print('C5 b$1 N $2-$3 append_ms %.4f step_ms %.4f frac %.3f' % (s['append_ms'], s['step_ms'], s['roofline']['frac']))"
    done
  done
done
