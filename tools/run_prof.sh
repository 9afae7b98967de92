cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # the setting bench.py gives itself; under rocprofv3 the runtime is up before Python runs
cd $GRAFT_REPO_ROOT
python bench.py --steps 200 --warmup 10 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_shared -- python3 bench.py --regime shared --steps 50 --warmup 5 --cpu-sample 0 > gpurun_out/bench_shared_prof.json 2> gpurun_out/bench_shared_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_indep -- python3 bench.py --steps 50 --warmup 5 --cpu-sample 0 > gpurun_out/bench_indep_prof.json 2> gpurun_out/bench_indep_prof.err
ls -R gpurun_out/prof_shared | head
tail -c 600 gpurun_out/bench_default.json
python bench.py --chunks 2 --cpu-sample 0 > gpurun_out/bench_chunks2.json 2>/dev/null
python bench.py --dtype f64 --cpu-sample 0 > gpurun_out/bench_f64.json 2>/dev/null
