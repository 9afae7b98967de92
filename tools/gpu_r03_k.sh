cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k
mkdir -p $O
timeout 900 python tools/bench_refit_forms.py f64 > $O/forms_f64.jsonl 2>$O/err.txt
timeout 900 python tools/bench_refit_forms.py f32 > $O/forms_f32.jsonl 2>>$O/err.txt
tail -2 $O/err.txt
python - <<'PY'
import json
for f in ("forms_f64","forms_f32"):
    for l in open("gpurun_out/r03k/%s.jsonl"%f):
        d=json.loads(l); print(f, d["batch"], d["N"], "wg %.3f wave %.3f pair %.3f" % (d["ms_workgroup"], d["ms_wave"], d["ms_pair"]), "fails", d["fail_pair"])
PY
