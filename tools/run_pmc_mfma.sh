# MFMA-utilisation counters for the matrix-core kernels (own pass: counters + kernel trace only)
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # the setting bench.py gives itself; under rocprofv3 the runtime is up before Python runs
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_shared -- python3 bench.py --regime shared --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/pmc_shared.log 2>&1
for cfg in C2 C3f64 C3 N1024f64; do      # one pass per config: workgroup counts coincide between configs (1024 / 4096)
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_refit_$cfg -- python3 tools/bench_configs.py $cfg > gpurun_out/pmc_refit_$cfg.log 2>&1
done
ls -R gpurun_out/pmc_shared | head; tail -2 gpurun_out/pmc_shared.log | cut -c1-300
