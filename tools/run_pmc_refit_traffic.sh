# Memory traffic of the refit kernels past L2 (FETCH_SIZE / WRITE_SIZE / L2 hits, one PMC pass each, counters + kernel trace
# only): config 2 (fp64 1024 x 256, two waves per instance) and config 3 (fp32 4096 x 512, one wave per instance).
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # the setting bench.py gives itself; under rocprofv3 the runtime is up before Python runs
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
for cfg in "f64 1024 256" "f32 4096 512"; do
  tag=$(echo $cfg | tr ' ' '_')
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d=$O/pmc_refit_traffic_${tag}_$(echo $c | tr ' ' '_')
    rm -rf $d
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/refit_only.py $cfg 6 > $d.log 2>&1
  done
done
python3 - <<PY
import csv, glob, json
out = {}
for tag, elems in (("f64_1024_256", 8), ("f32_4096_512", 4)):
    acc, dur = {}, []
    for f in glob.glob("$O/pmc_refit_traffic_%s_*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "refit_" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                kern = r["Kernel_Name"].split("(")[0]
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    # gfx950: FETCH_SIZE counts 2 KiB units... as MI355X_MICROARCH.md prescribes: bytes = FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024
    out[tag] = dict(kernel=kern, launches=len(next(iter(acc.values()))), fetch_bytes=m.get("FETCH_SIZE", 0) * 2048, write_bytes=m.get("WRITE_SIZE", 0) * 1024,
                    l2_hit=m.get("TCC_HIT_sum"), l2_miss=m.get("TCC_MISS_sum"),
                    l2_hit_rate=(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])) if "TCC_HIT_sum" in m else None)
json.dump(out, open("$O/pmc_traffic_refit.json", "w"), indent=1)
print(json.dumps(out))
PY
