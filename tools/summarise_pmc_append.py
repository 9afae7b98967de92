"""Summary of tools/run_pmc_append_traffic.sh: per configuration the HBM bytes the online path's two kernels moved (sum over
all their launches; gfx950 units as MI355X_MICROARCH.md prescribes: FETCH_SIZE x 2 KiB ... see run_pmc_refit_traffic.sh,
WRITE_SIZE x 1 KiB) against the algorithmic bytes of the same appends, and the roofline fraction from the kernel trace's
own durations.  Writes <dir>/pmc_traffic_append.json."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_cbf_amd.rollouts import online_pass_bytes

O = sys.argv[1]
out = {}
for tag, (Bt, n0, n1) in {"b256_n1024_2048": (256, 1024, 2048), "b1024_n1024_1280": (1024, 1024, 1280)}.items():
    alg = sum(online_pass_bytes(N, 3, 2, 8) for N in range(n0, n1)) * Bt
    rec = dict(batch=Bt, N_from=n0, N_to=n1, dtype="f64", appends=n1 - n0, algorithmic_bytes_total=alg,
               algorithmic_bytes_per_append=alg / (n1 - n0))
    per_kernel = {}
    for f in glob.glob(os.path.join(O, "pmc_append_%s_*" % tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "posterior_step_kernel" in k and ", false, 1" in k.split("(")[0]:
                name = k.split("(")[0].replace("void bcbf::", "")
            elif "gp_append_inplace_kernel" in k:
                name = "gp_append_inplace_kernel<double>"
            else:
                continue
            d = per_kernel.setdefault(name, {})
            d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    tot_f = tot_w = 0.0
    for name, c in per_kernel.items():
        fb = sum(c.get("FETCH_SIZE", [])) * 2048
        wb = sum(c.get("WRITE_SIZE", [])) * 1024
        hit, miss = sum(c.get("TCC_HIT_sum", [])), sum(c.get("TCC_MISS_sum", []))
        rec[name] = dict(launches=len(c.get("FETCH_SIZE", [])), fetch_bytes=fb, write_bytes=wb,
                         l2_hit_rate=hit / (hit + miss) if hit + miss else None)
        tot_f += fb
        tot_w += wb
    rec["hbm_bytes_total"] = tot_f + tot_w
    rec["traffic_over_algorithmic"] = (tot_f + tot_w) / alg if alg else None
    # kernel trace durations (the --stats pass): total ns of the two kernels
    dur = {}
    for f in glob.glob(os.path.join(O, "prof_append_%s" % tag, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Name"]
            if ("posterior_step_kernel" in k and ", false, 1" in k.split("(")[0]) or "gp_append_inplace_kernel" in k:
                dur[k.split("(")[0]] = dict(calls=int(r["Calls"]), total_ns=float(r["TotalDurationNs"]), avg_ns=float(r["AverageNs"]))
    if dur:
        tot = sum(v["total_ns"] for v in dur.values())
        gbs = alg / (tot * 1e-9) / 1e9
        rec["kernel_trace"] = dict(kernels=dur, achieved_GBs=gbs, peak_GBs=8000.0, frac=gbs / 8000.0,
                                   how="algorithmic bytes of all appends / summed trace durations of the two kernels")
    out[tag] = rec
json.dump(out, open(os.path.join(O, "pmc_traffic_append.json"), "w"), indent=1)
print(json.dumps(out))
