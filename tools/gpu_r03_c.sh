cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c
mkdir -p $O
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "reserved or c5 or append or c3" 2>&1 | tail -8 > $O/pytest_sel.txt
python tools/bench_online.py --repeat 3 > $O/online_reserved.jsonl 2>$O/online.err
python tools/bench_online.py --packed > $O/online_packed.jsonl 2>>$O/online.err
python tools/bench_reldeg2.py > $O/reldeg2.jsonl 2>$O/reldeg2.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>$O/bench_driver_form.err
python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_driver_form2.json 2>>$O/bench_driver_form.err
python bench.py --cpu-sample 0 > $O/bench_default.json 2>$O/bench_default.err
cat $O/pytest_sel.txt; tail -3 $O/online.err
python - <<'PY'
import json
for f in ("online_reserved","online_packed"):
    for l in open("gpurun_out/r03c/%s.jsonl"%f):
        d=json.loads(l); print(f, [(s["N_from"], round(s["append_ms"],3), round(s["control_step_ms"],3)) for s in d["segments"]], d["append_failures"], d.get("final_vs_refit"))
for l in open("gpurun_out/r03c/reldeg2.jsonl"):
    d=json.loads(l); print("reldeg2", d["n"], d["m"], d["dtype"], round(d["jets_ms"],4), round(d["jets_GBs_algorithmic"]), round(d["values_only_ms"],4), round(d["cbc2_terms_ms"],4))
for f in ("bench_driver_form","bench_driver_form2","bench_default"):
    try:
        d=json.loads([l for l in open("gpurun_out/r03c/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["value"]), round(d["ms_per_step"],4), d["warmup"], d["steps"], d["timed_region"], round(d["roofline"]["frac"],4))
    except Exception as e: print(f, "ERR", e)
PY
