# round 3, first GPU call: full GPU suite, jets-kernel variants, bench lines (driver form + default), kernel trace + union
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
python tools/tune_jets.py run > $O/tune_jets_f32.txt 2>&1
python tools/tune_jets.py run f64 > $O/tune_jets_f64.txt 2>&1
python tools/bench_reldeg2.py > $O/reldeg2.jsonl 2>$O/reldeg2.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>$O/bench_driver_form.err
python bench.py > $O/bench_default.json 2>$O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 bench.py --steps 50 --warmup 5 --cpu-sample 0 > $O/bench_default_prof.json 2> $O/bench_default_prof.err
python tools/trace_union.py $O/prof_default $O/bench_default_prof.json $O/bench_default_prof_union.json > $O/union.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_kernel_trace.csv" -size +3M -delete 2>/dev/null
cat $O/pytest.txt; tail -12 $O/tune_jets_f32.txt; tail -12 $O/tune_jets_f64.txt; cat $O/reldeg2.jsonl; cat $O/union.txt
python - <<'PY'
import json
for f in ("bench_driver_form","bench_default","bench_default_prof"):
    try:
        d=json.loads([l for l in open("gpurun_out/r03a/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["value"]), round(d["ms_per_step"],4), d["warmup"], d["steps"], d["timed_region"], round(d["roofline"]["frac"],4), d.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
