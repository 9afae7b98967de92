cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f
mkdir -p $O
python -m pytest tests/test_gpu_facade.py tests/test_gpu_bench.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt
python bench.py --cpu-sample 0 > $O/bench_p3.json 2>$O/p3.err
python bench.py --cpu-sample 0 --steps 20 --warmup 5 > $O/bench_p3_driver.json 2>/dev/null
for q in 4 5 6; do GPU_MAX_HW_QUEUES=8 python bench.py --cpu-sample 0 --parts $q > $O/bench_q8_p$q.json 2>/dev/null; done
GPU_MAX_HW_QUEUES=8 python bench.py --cpu-sample 0 --parts 3 > $O/bench_q8_p3.json 2>/dev/null
python bench.py --cpu-sample 0 --dtype f64 > $O/bench_p3_f64.json 2>/dev/null
python bench.py --cpu-sample 0 --dtype f64 --parts 2 > $O/bench_p2_f64.json 2>/dev/null
python bench.py --cpu-sample 0 --regime shared > $O/bench_p3_shared.json 2>/dev/null
python bench.py --cpu-sample 0 --regime shared --parts 2 > $O/bench_p2_shared.json 2>/dev/null
cat $O/pytest.txt; tail -3 $O/p3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03f/bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), "busy", round(d["roofline"]["kernel_busy_ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), d["timed_region"]["first_steps_ms"])
    except Exception as e: print(f, "ERR", e)
PY
