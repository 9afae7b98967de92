# A/B of libbcbf variants on the C5 growth bench (development): append + fused query, fp64, N 1024 -> 2048
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "reserved or c5 or window or learning or online" 2>&1 | tail -5 > gpurun_out/ab_c5_tests.log
for round in 1 2; do
  for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_*.so; do
    echo "== $lib" >> gpurun_out/ab_c5.log
    BCBF_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_online.py --n0 1024 --n1 2048 2>&1 | grep -v amdgpu.ids | cut -c1-600 >> gpurun_out/ab_c5.log
  done
done
