cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest.txt
python tools/bench_online.py --repeat 2 > $O/online_fused.jsonl 2>$O/online.err
python tools/bench_online.py --unfused > $O/online_unfused.jsonl 2>>$O/online.err
python tools/bench_online.py --dtype f32 > $O/online_fused_f32.jsonl 2>>$O/online.err
cat $O/pytest.txt; tail -3 $O/online.err
python - <<'PY'
import json
for f in ("online_fused","online_unfused","online_fused_f32"):
    for l in open("gpurun_out/r03g/%s.jsonl"%f):
        d=json.loads(l); print(f, [(s["N_from"], round(s["append_ms"],3), round(s["control_step_ms"],3), round(s.get("step_ms",0),3)) for s in d["segments"]], d["append_failures"], d.get("final_vs_refit"))
PY
