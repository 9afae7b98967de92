cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc_refit_C2 -- python3 tools/bench_configs.py C2 > $O/pmc_refit_C2.log 2>&1
cat $O/pytest.txt
python - <<'PY'
import json, csv, glob, collections
for l in open("gpurun_out/r03l/configs.jsonl"):
    d=json.loads(l); print(d["config"], "refit_ms", round(d["refit_ms"],3), "TF", round(d["refit_TFLOPs"],1))
f=glob.glob("gpurun_out/r03l/pmc_refit_C2/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "refit" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    m={a:sum(v)/len(v) for a,v in c.items()}
    print(k, "mfma_util", round(m["SQ_VALU_MFMA_BUSY_CYCLES"]/(m["GRBM_GUI_ACTIVE"]/8*1024),4), "wait", round(m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"],3), "launches", len(c["SQ_WAVE_CYCLES"]))
PY
