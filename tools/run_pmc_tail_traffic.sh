# HBM traffic of the online_tail schedule's two kernels -- the streaming pass over the static window
# (`posterior_step_kernel<float, 3, 4, 0, 1, false, 1>`) and `gp_tail_step_kernel` -- at 4096 x (472 + t <= 39), fp32:
# two PMC passes (counters + kernel trace only, the program directly after `--`), summary -> gpurun_out/$R/pmc_traffic_tail.json
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/pmc_tail_$c
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/bench_learning_loop.py --schedule online_tail --steps 80 --warmup 40 > $d.log 2>&1
done
python3 - <<PY
import csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
from bayesian_cbf_amd.rollouts import online_pass_bytes
O = "$O"
per = {}
dur = {}
for f in glob.glob(os.path.join(O, "pmc_tail_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "posterior_step_kernel<float, 3, 4, 0, 1, false, 1>" if ("posterior_step_kernel" in k and ", false, 1" in k.split("(")[0]) else ("gp_tail_step_kernel<float, 4>" if "gp_tail_step_kernel" in k else None)
        if name:
            per.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(O, "pmc_tail_FETCH_SIZE", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "posterior_step_kernel<float, 3, 4, 0, 1, false, 1>" if ("posterior_step_kernel" in k and ", false, 1" in k.split("(")[0]) else ("gp_tail_step_kernel<float, 4>" if "gp_tail_step_kernel" in k else None)
        if name:
            dur.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
Bt, N0 = 4096, 472
alg_pass = online_pass_bytes(N0, 3, 2, 4) * Bt                    # the window's triangle + arrays, read once
out = dict(batch=Bt, window=N0, dtype="f32", algorithmic_bytes_per_step_at_N_live="Bt x online_pass_bytes(472 + t)", kernels={})
for name, c in per.items():
    n = len(c.get("FETCH_SIZE", []))
    fb = sum(c.get("FETCH_SIZE", [])) * 2048 / max(1, n)
    wb = sum(c.get("WRITE_SIZE", [])) * 1024 / max(1, len(c.get("WRITE_SIZE", [])))
    ns = sum(dur.get(name, [0])) / max(1, len(dur.get(name, [])))
    out["kernels"][name] = dict(launches=n, fetch_bytes_per_launch=fb, write_bytes_per_launch=wb, avg_ns_under_counters=ns)
p = out["kernels"].get("posterior_step_kernel<float, 3, 4, 0, 1, false, 1>")
if p:
    p["algorithmic_bytes_per_launch"] = alg_pass
    p["traffic_over_algorithmic"] = (p["fetch_bytes_per_launch"] + p["write_bytes_per_launch"]) / alg_pass
t = out["kernels"].get("gp_tail_step_kernel<float, 4>")
if t:
    tail_alg = Bt * 4 * (19.5 * N0 + 480 * 4 + N0)               # mean t = 19.5 rows of N0, the solved columns [480, 4], the new row
    t["algorithmic_bytes_per_launch_mean_t"] = tail_alg
    t["traffic_over_algorithmic"] = (t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]) / tail_alg
json.dump(out, open(os.path.join(O, "pmc_traffic_tail.json"), "w"), indent=1)
print(json.dumps(out)[:900])
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -path "*pmc_tail_*" -name "*_kernel_trace.csv" -delete 2>/dev/null; find $O -path "*pmc_tail_*" -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
