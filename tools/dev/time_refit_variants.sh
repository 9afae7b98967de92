# tools/dev/time_refit_variants.sh: the batched fp32 refit (4096 x 512, 1024 x 512, 1024 x 1024) for every library under
# tools/_variants plus the product library, with both register allocations of the one-wave form forced (development).
cd $GRAFT_REPO_ROOT
for lib in bayesian_cbf_amd/libbcbf.so tools/_variants/libbcbf_*.so; do
  for occ in 1 2; do
    echo "== $lib occ=$occ"
    BCBF_LIB_PATH=$lib BCBF_RW32_OCC=$occ BCBF_REFIT_WAVE=1 python tools/dev/time_refit32.py 2>&1 | tail -3
  done
done
