# HBM traffic of the roofline kernel: three PMC passes (counters + kernel trace only), as MI355X_MICROARCH.md prescribes
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # the setting bench.py gives itself; under rocprofv3 the runtime is up before Python runs
cd $GRAFT_REPO_ROOT
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  d=gpurun_out/pmc_traffic_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 > $d.log 2>&1
done
ls gpurun_out | grep pmc_traffic
