# development: counters of the fused kernel in the online loop against the same call in isolation (tools/dev/diag_clock.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/diag_clock
mkdir -p $O
for c in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  d=$O/$(echo $c | tr ' ' '_')
  rm -rf $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/dev/diag_clock.py > $d.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
O = "gpurun_out/diag_clock"
for d in sorted(glob.glob(O + "/*/")):
    cc = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "**/*kernel_trace.csv", recursive=True)
    if not cc or not kt:
        print(d, "no csv", cc, kt); continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    rows = {}
    for r in csv.DictReader(open(cc[0])):
        k = r["Kernel_Name"]
        if "posterior_step_kernel" in k and ", false, 1" in k.split("(")[0]:
            rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows, key=int)
    print(d, "launches", len(ids))
    def show(tag, sel):
        if not sel: return
        ns = sum(dur[i][0] for i in sel if i in dur) / len(sel)
        cs = {c: sum(rows[i].get(c, 0) for i in sel) / len(sel) for c in rows[sel[0]]}
        print("  %-10s n=%d avg_ns %.0f  " % (tag, len(sel), ns) + "  ".join("%s %.4g (per ns %.4g)" % (c, v, v / ns) for c, v in cs.items()))
    show("in loop", ids[40:80])          # the schedule's second window (the first follows the initial fit)
    show("isolated", ids[80 + 39:])      # the plain loop's second and third rounds
PY
