#!/usr/bin/env python3
"""Copy the summaries of the last tools/run_prof.sh / run_pmc_mfma.sh / bench runs from gpurun_out/ into profiles/."""
import collections, csv, glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
latest = lambda pat: sorted(glob.glob(R + pat), key=os.path.getmtime)[-1]
for tag, dst in (("prof_indep", "profiles/r01_bench_kernel_stats.csv"), ("prof_shared", "profiles/r01_bench_shared_kernel_stats.csv")):
    rows = list(csv.reader(open(latest("gpurun_out/%s/*/*_kernel_stats.csv" % tag))))
    out = [rows[0]] + [r for r in rows[1:] if "bcbf::" in r[0]]
    csv.writer(open(R + dst, "w")).writerows(out)
    print(dst, [(r[0].split("(")[0][-42:], r[1], round(float(r[3]) / 1e3, 1)) for r in out[1:]])
for a, b in (("bench_default.json", "r01_bench_default.json"), ("bench_shared_prof.json", "r01_bench_shared_under_rocprof.json"),
             ("bench_shared.json", "r01_bench_shared.json"), ("configs.jsonl", "r01_configs_refit_potrs_posterior.jsonl"),
             ("online_growth_f64.json", "r01_online_growth_f64.json"), ("speed_test.jsonl", "r01_speed_test_matrix_vector.jsonl"),
             ("learn_matrix_vector.jsonl", "r01_learn_dynamics_matrix_vector.jsonl")):
    if os.path.exists(R + "gpurun_out/" + a):
        shutil.copy(R + "gpurun_out/" + a, R + "profiles/" + b)
d = json.load(open(R + "profiles/r01_bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
d = json.load(open(R + "profiles/r01_bench_shared.json"))
print("shared", d["value"], d["batched_steps_per_s"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
out = {}
for tag, pat in (("bench_shared", "gpurun_out/pmc_shared/*/*_counter_collection.csv"),
                 ("bench_configs C2", "gpurun_out/pmc_refit_C2/*/*_counter_collection.csv"),
                 ("bench_configs C3f64", "gpurun_out/pmc_refit_C3f64/*/*_counter_collection.csv"),
                 ("bench_configs C3", "gpurun_out/pmc_refit_C3/*/*_counter_collection.csv"),
                 ("bench_configs N1024f64", "gpurun_out/pmc_refit_N1024f64/*/*_counter_collection.csv")):
    if not glob.glob(R + pat):
        continue                       # no fresh counter pass (tools/run_pmc_mfma.sh): keep the committed numbers
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(latest(pat))):
        k = r["Kernel_Name"]
        if "mfma" not in k and "posterior_shared" not in k:
            continue
        agg[(k.split("(")[0].replace("void bcbf::", ""), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (name, wgs), c in sorted(agg.items()):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        if wgs < 64:
            continue
        util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        dd = dict(kernel=name, workgroups=wgs, launches=len(next(iter(c.values()))), gpu_cycles=m["GRBM_GUI_ACTIVE"] / 8,
                  mfma_busy_cycles_all_simds=m["SQ_VALU_MFMA_BUSY_CYCLES"], mfma_util=round(util, 4),
                  wait_any_frac=round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3), wait_inst_frac=round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3),
                  active_inst_frac=round(m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3))
        out.setdefault(tag, []).append(dd)
        print(tag, dd["kernel"], dd["workgroups"], dd["mfma_util"])
old = json.load(open(R + "profiles/r01_pmc_mfma.json"))
if out:
    json.dump(dict(note=old["note"], passes=out), open(R + "profiles/r01_pmc_mfma.json", "w"), indent=1)
for l in open(R + "profiles/r01_configs_refit_potrs_posterior.jsonl"):
    d = json.loads(l)
    print(d["config"], round(d["refit_ms"], 3), round(d["refit_TFLOPs"], 1), round(d["potrs_ms"], 3), round(d["posterior_ms"], 4), round(d["posterior_GBs_algorithmic"]))
for l in open(R + "profiles/r01_speed_test_matrix_vector.jsonl"):
    d = json.loads(l)
    print(d["regressor"], d["N"], round(d["s_per_call"] * 1e3, 2), "ms", round(d["speedup_vs_published"], 1), round(d["fit_s"], 2), round(d["heldout_rel_rms_err"], 4))
for l in open(R + "profiles/r01_learn_dynamics_matrix_vector.jsonl"):
    d = json.loads(l)
    print("learn", d["N_train"], d["seed"], round(d["seconds"], 2), "s  err matrix/vector", round(d["error_matrix"], 3), round(d["error_vector"], 3))
