# tools/dev/pmc_refit_variant.sh <lib> <occ> <tag>: FETCH_SIZE / WRITE_SIZE / MFMA-busy of the fp32 one-wave refit at 4096 x 512 for one
# library (development; one counter set per pass, counters + kernel trace only)
LIB=$1; OCC=$2; TAG=$3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export BCBF_LIB_PATH=$LIB BCBF_RW32_OCC=$OCC BCBF_REFIT_WAVE=1
O=gpurun_out/pmcv_$TAG
mkdir -p $O
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$O/$(echo $c | tr ' ' '_')
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/refit_only.py f32 4096 512 6 > $d.log 2>&1
done
python3 - <<PY
import csv, glob, json
out = {}
for f in glob.glob("$O/*/*/*counter_collection.csv"):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "refit_" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
dur = []
for f in glob.glob("$O/FETCH_SIZE/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "refit_" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
res = dict(tag="$TAG", fetch_GB=out.get("FETCH_SIZE", 0) * 2048 / 1e9, write_GB=out.get("WRITE_SIZE", 0) * 1024 / 1e9,
           mfma_busy=out.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(out.get("SQ_BUSY_CYCLES", 1), 1), ms=sorted(dur)[len(dur) // 2] if dur else None,
           raw=out)
print(json.dumps(res))
PY
