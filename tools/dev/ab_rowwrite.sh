# A/B (development): the new operator row written by the fused pass (BCBF_PX_ROWWRITE) against the row kernel's scattered writes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "reserved or c5 or window or learning or online or append" 2>&1 | tail -5 > gpurun_out/ab_rw_tests.log
for round in 1 2; do
  for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_*.so; do
    echo "== $lib" >> gpurun_out/ab_rw.log
    BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_online.py --n0 1024 --n1 2048 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['segments'][0]
print('C5 b256 append_ms %.4f step_ms %.4f frac %.3f' % (s['append_ms'], s['step_ms'], s['roofline']['frac']))" >> gpurun_out/ab_rw.log
    BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_online.py --batch 1024 --n0 1024 --n1 1280 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['segments'][0]
print('C5 b1024 append_ms %.4f step_ms %.4f frac %.3f' % (s['append_ms'], s['step_ms'], s['roofline']['frac']))" >> gpurun_out/ab_rw.log
    BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_learning_loop.py --steps 200 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('learn online pass_ms %.4f  ms_per_step %.4f  value %.3f M/s' % (d['shares']['pass_ms_per_step'], d['ms_per_step'], d['value'] / 1e6))" >> gpurun_out/ab_rw.log
  done
done
