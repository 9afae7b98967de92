# A/B of the jets kernel variants under tools/_variants (development): the rel-degree-2 bench per library (unicycle shape line)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in 1 2; do
  for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_*.so; do
    echo "== $lib" >> gpurun_out/ab_jets.log
    BCBF_LIB_PATH=$PWD/$lib timeout 150 python tools/bench_reldeg2.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> gpurun_out/ab_jets.log
    echo "rc=$?" >> gpurun_out/ab_jets.log
  done
done
