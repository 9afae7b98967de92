cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stag
run() { python bench.py $2 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['steps'], d['warmup'], round(d['ms_per_step'],4), d['timed_region']['first_steps_ms'], d['timed_region']['last_steps_ms'], round(d['roofline']['frac'],3))"; }
BCBF_BENCH_DUMP=gpurun_out/stag/e1.json run plain "--steps 200 --warmup 60"
BCBF_BENCH_NOSYNC=1 BCBF_BENCH_DUMP=gpurun_out/stag/e2.json run nosync "--steps 200 --warmup 60"
BCBF_BENCH_WEV=1 BCBF_BENCH_DUMP=gpurun_out/stag/e3.json run wev "--steps 200 --warmup 60"
BCBF_BENCH_WEV=1 BCBF_BENCH_NOSYNC=1 BCBF_BENCH_DUMP=gpurun_out/stag/e4.json run wev_nosync "--steps 200 --warmup 60"
BCBF_BENCH_DUMP=gpurun_out/stag/e5.json run parts1 "--steps 200 --warmup 60 --parts 1"
BCBF_BENCH_DUMP=gpurun_out/stag/e6.json run parts2 "--steps 200 --warmup 60 --parts 2"
