cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
python tools/prof_fit.py > $O/prof_fit.txt 2>&1
for i in 1 2; do
python bench.py --cpu-sample 0 > $O/bench_base_$i.json 2>/dev/null
BCBF_LIB_PATH=$PWD/tools/_variants/libbcbf_prio.so python bench.py --cpu-sample 0 > $O/bench_prio_$i.json 2>/dev/null
done
python bench.py --cpu-sample 0 --batch 4095 --parts 3 > $O/bench_parts3.json 2>$O/parts3.err
python bench.py --cpu-sample 0 --batch 4095 --parts 1 > $O/bench_4095_parts1.json 2>/dev/null
python bench.py --cpu-sample 0 --batch 4096 --parts 4 > $O/bench_parts4.json 2>/dev/null
cat $O/pytest.txt; grep "fit 50" $O/prof_fit.txt; tail -3 $O/parts3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03e/bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), "busy", round(d["roofline"]["kernel_busy_ms_per_step"],4), "frac", round(d["roofline"]["frac"],4))
    except Exception as e: print(f, "ERR", e)
PY
