cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j
mkdir -p $O
BCBF_REFIT_PAIR=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "refit or c2 or c3 or posterior_vs or golden" 2>&1 | tail -8 > $O/pytest_pair.txt
timeout 300 python tools/bench_configs.py > $O/configs_base.jsonl 2>/dev/null
BCBF_REFIT_PAIR=1 timeout 300 python tools/bench_configs.py > $O/configs_pair.jsonl 2>/dev/null
cat $O/pytest_pair.txt
python - <<'PY'
import json
for f in ("configs_base","configs_pair"):
    for l in open("gpurun_out/r03j/%s.jsonl"%f):
        d=json.loads(l); print(f, d["config"], "refit_ms", round(d["refit_ms"],3), "TF", round(d["refit_TFLOPs"],1))
PY
