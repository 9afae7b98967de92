cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h
mkdir -p $O
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s -k "timed_schedule" 2>&1 | grep -E "C3 two-stream|passed|failed" > $O/bk.txt
for P in 2 3 4; do GPU_MAX_HW_QUEUES=8 python bench.py --cpu-sample 0 --regime shared --parts $P > $O/shared_p$P.json 2>/dev/null; GPU_MAX_HW_QUEUES=8 python bench.py --cpu-sample 0 --regime shared --dtype f64 --parts $P > $O/shared64_p$P.json 2>/dev/null; done
python bench.py --cpu-sample 0 --regime shared --parts 1 > $O/shared_p1.json 2>/dev/null
cat $O/bk.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03h/shared*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), "busy", round(d["roofline"]["kernel_busy_ms_per_step"],4), "frac", round(d["roofline"]["frac"],4))
    except Exception as e: print(f, "ERR", e)
PY
