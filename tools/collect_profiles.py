#!/usr/bin/env python3
"""Copy the summaries of the last `tools/run_profiles.sh <round>` run from gpurun_out/<round>/ into profiles/<round>_*.

    python tools/collect_profiles.py r02
"""
import collections, csv, glob, json, os, shutil, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = R + "gpurun_out/%s/" % ROUND
DST = R + "profiles/%s_" % ROUND


def latest(pat):
    hits = sorted(glob.glob(SRC + pat), key=os.path.getmtime)
    return hits[-1] if hits else None


def last_json_line(path):
    lines = [l for l in open(path) if l.startswith("{")]
    return json.loads(lines[-1])


# ---- kernel-trace summaries (libbcbf kernels only; the torch kernels of the synthetic-data generator are dropped)
for tag in ("default", "parts1", "shared", "shared_f64", "reldeg2", "learn_reference", "learn_online", "learn_online_tail", "learn_loop_reference", "fit_iter", "speed_call", "append_b256_n1024_2048", "append_b1024_n1024_1280"):
    f = latest("prof_%s/*/*_kernel_stats.csv" % tag)
    if not f:
        continue
    rows = list(csv.reader(open(f)))
    own = [r for r in rows[1:] if "bcbf::" in r[0]]
    # everything that is NOT a libbcbf kernel (torch's elementwise / copy / index launches, a library GEMM of the synthetic-data
    # generator): one summed row, so that whatever the host path launches beside the library shows up in the share column
    rest = [r for r in rows[1:] if "bcbf::" not in r[0]]
    other = []
    if rest:
        calls, tot = sum(int(r[1]) for r in rest), sum(float(r[2]) for r in rest)
        biggest = max(rest, key=lambda r: float(r[2]))
        other = [["torch / other (%d kernels; largest: %s)" % (len(rest), biggest[0][:80]), calls, int(tot), tot / max(1, calls),
                  sum(float(r[4]) for r in rest), min(float(r[5]) for r in rest), max(float(r[6]) for r in rest), ""]]
    out = [rows[0]] + own + other
    csv.writer(open(DST + "bench_%s_kernel_stats.csv" % tag, "w")).writerows(out)
    print(tag, [(r[0].split("(")[0][-44:], r[1], round(float(r[3]) / 1e3, 1)) for r in out[1:]])

# ---- bench lines
for a in ("bench_default", "bench_driver_form", "bench_default_prof", "bench_parts1", "bench_parts1_prof", "bench_parts2", "bench_parts3", "bench_shared", "bench_shared_prof", "bench_shared_f64", "bench_shared_f64_prof", "bench_f64", "bench_shared_b10240", "bench_shared_b10240_f64"):
    if os.path.exists(SRC + a + ".json") and os.path.getsize(SRC + a + ".json"):
        d = last_json_line(SRC + a + ".json")
        json.dump(d, open(DST + a + ".json", "w"), indent=1)
        rf = d["roofline"]
        print(a, round(d["value"]), round(d["ms_per_step"], 4), "frac", round(rf["frac"], 4), "kernel_ms", round(rf["kernel_ms"], 4),
              "busy/step", round(rf.get("kernel_busy_ms_per_step", 0), 4), d.get("cpu_baseline", {}).get("value"))
for a in ("configs.jsonl", "refit_forms.jsonl", "refit_forms_f32.jsonl", "online_growth_f64.json", "online_growth_f64_unfused.json", "online_growth_f64_unfused3.json", "online_growth_f64_packed.json", "online_growth_f64_batch1024.json", "online_growth_f64_pairform.json", "online_window512_f64.json", "online_window1024_f64.json", "speed_call_host.txt", "reldeg2.jsonl", "speed_test.jsonl", "speed_test_unicycle.jsonl",
          "learn_matrix_vector.jsonl", "mc_rollouts.txt", "shared_queries.txt", "shared_sweep.jsonl", "bench_default_prof_union.json", "bench_parts1_prof_union.json", "ramp.txt", "pmc_traffic_refit.json",
          "refit_pair_timeline.txt", "learn_reference_parts4.json", "learn_reference.json", "learn_online.json", "learn_online_tail.json", "learn_online_tail_prof.json", "learn_reference_prof.json",
          "learn_online_prof.json", "pmc_traffic_append.json", "online_b256_n1024_2048.json", "online_b1024_n1024_1280.json",
          "online_f32_b4096.json", "tol_report.txt", "learn_loop_reference_f32.json", "learn_loop_reference_f64.json",
          "learn_loop_reference_nostagger_f32.json", "learn_loop_reference_nostagger_f64.json", "learn_loop_online_tail_f32.json",
          "learn_loop_online_tail_f64.json", "learn_loop_reference_fit100_f32.json", "learn_loop_reference_fit100_f32_b256.json",
          "learn_loop_reference_prof.json", "learn_loop_reference_mixed.json", "learn_loop_reference_f32_floor1e-3.json", "learn_loop_reference_f64_floor1e-3.json", "learn_loop_reference_fit100_mixed_b256.json", "trtri_syrk.jsonl", "fit_iteration.jsonl", "refit_footprint.jsonl", "online_growth_tail.jsonl"):
    if os.path.exists(SRC + a) and os.path.getsize(SRC + a):
        shutil.copy(SRC + a, DST + a)

# ---- HBM traffic of the roofline kernel from the three counter passes per schedule
NOTE = ("FETCH_SIZE is in KiB and, on gfx950, counts 1/2 of the bytes of wide coalesced reads: bytes = FETCH_SIZE*1024*2; "
        "WRITE_SIZE*1024 exact.  Cross-check: TCC_MISS_sum * 128 B.  (MI355X_MICROARCH.md, HBM / rocprofv3 section)")
def source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in ("bayesian_cbf_amd/csrc/posterior_step.hip", "bayesian_cbf_amd/csrc/bcbf_common.h"):
        h.update(open(R + f, "rb").read())
    return h.hexdigest()


def traffic_pass(tag, kernel_sub, per_instance_alg, command, out_name, sizes=None, dtype="f32", n=3, m=2):
    """Per-instance HBM traffic of one kernel from the three counter passes pmc_traffic_<tag>_{FETCH_SIZE,WRITE_SIZE,TCC..}."""
    per_inst = collections.defaultdict(list)
    launches = collections.defaultdict(list)
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
        f = latest("pmc_traffic_%s_%s/*/*_counter_collection.csv" % (tag, c))
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            wgs = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
            if kernel_sub in r["Kernel_Name"] and (sizes is None or wgs in sizes):
                per_inst[r["Counter_Name"]].append(float(r["Counter_Value"]) / wgs)
                launches[r["Counter_Name"]].append(wgs)
    if "FETCH_SIZE" not in per_inst or "WRITE_SIZE" not in per_inst:
        return
    mean = lambda k: sum(per_inst[k]) / len(per_inst[k])
    fetch, write = mean("FETCH_SIZE") * 1024 * 2, mean("WRITE_SIZE") * 1024
    batch = sum(launches["FETCH_SIZE"]) / len(launches["FETCH_SIZE"])
    out = dict(kernel=kernel_sub, workload=dict(N_train=512, batch=batch, dtype=dtype, n=n, m=m, schedule=command,
                                                workgroups_per_launch=sorted(set(launches["FETCH_SIZE"]))),
               commands=["rocprofv3 --pmc %s --kernel-trace --output-format csv -- python3 %s" % (c, command)
                         for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum")],
               launches=len(per_inst["FETCH_SIZE"]), corrections=NOTE, fetch_bytes_per_instance=fetch, write_bytes_per_instance=write,
               hbm_bytes_per_instance=fetch + write, hbm_bytes_per_launch=(fetch + write) * batch,
               tcc_miss_bytes_per_instance=(mean("TCC_MISS_sum") * 128 if "TCC_MISS_sum" in per_inst else None),
               algorithmic_bytes_per_instance=per_instance_alg, algorithmic_bytes_per_launch=per_instance_alg * batch,
               traffic_over_algorithmic=(fetch + write) / per_instance_alg,
               # the kernel source these counters were taken from: bench.py attaches `traffic` only while the file still hashes to this
               kernel_source=dict(file="bayesian_cbf_amd/csrc/posterior_step.hip", sha256=source_hash()))
    json.dump(out, open(DST + out_name, "w"), indent=1)
    print("traffic", tag, kernel_sub, round(fetch + write), "x algorithmic", round(out["traffic_over_algorithmic"], 4), "launches", out["launches"])


HEAD = "posterior_step_kernel<float, 3, 4, 0, 1, false"        # (+ the XC parameter from round 4 on)
sizes_default = None
try:
    sizes_default = set(last_json_line(SRC + "bench_default.json")["roofline"]["instances_per_launch_by_part"])
except Exception:
    pass
traffic_pass("default", HEAD, 543744, "bench.py --steps 5 --warmup 2 --cpu-sample 0", "pmc_traffic.json", sizes_default)
traffic_pass("defaultparts1", HEAD, 543744, "bench.py --steps 5 --warmup 2 --cpu-sample 0 --parts 1", "pmc_traffic_parts1.json", {4096})
# rel-degree-2 jets (tools/bench_reldeg2.py): the unicycle shape (12 right-hand sides) and the pendulum shape (6)
traffic_pass("jets", "posterior_jets_mfma_kernel<3, 3>", 543744, "tools/bench_reldeg2.py", "pmc_traffic_jets_n3m2.json", {4096})      # (round 6: twelve columns run jets_mfma.hip)
traffic_pass("jets", "posterior_step_kernel<float, 2, 4, 2, 1, false", 4 * (512 * 513 // 2 + 2 * 512 * 2 + 512 * 2), "tools/bench_reldeg2.py",
             "pmc_traffic_jets_n2m1.json", {4096}, n=2, m=1)

# ---- MFMA utilisation
passes = {}
for tag, sub in (("bench.py --regime shared --parts 1", "pmc_shared"), ("bench.py --regime shared --dtype f64 --parts 1", "pmc_shared_f64"), ("bench_configs C2", "pmc_refit_C2"),
                 ("bench_configs C3f64", "pmc_refit_C3f64"), ("bench_configs C3", "pmc_refit_C3"),
                 ("bench_configs N1024f64", "pmc_refit_N1024f64")):
    f = latest(sub + "/*/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "refit" not in k and "posterior_shared" not in k:
            continue
        agg[(k.split("(")[0].replace("void bcbf::", ""), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (name, wgs), c in sorted(agg.items()):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        if wgs < 64 or "SQ_VALU_MFMA_BUSY_CYCLES" not in m:
            continue
        util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        dd = dict(kernel=name, workgroups=wgs, launches=len(next(iter(c.values()))), gpu_cycles=m["GRBM_GUI_ACTIVE"] / 8,
                  mfma_busy_cycles_all_simds=m["SQ_VALU_MFMA_BUSY_CYCLES"], mfma_util=round(util, 4),
                  wait_any_frac=round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3), wait_inst_frac=round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3),
                  active_inst_frac=round(m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3))
        passes.setdefault(tag, []).append(dd)
        print("mfma", tag, dd["kernel"], dd["workgroups"], dd["mfma_util"], "wait", dd["wait_any_frac"])
if passes:
    note = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
            "GRBM_GUI_ACTIVE --kernel-trace (own pass per command).  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs "
            "* 1024 SIMDs): fraction of SIMD-cycles with the MFMA pipe busy, averaged over the launches of the kernel.")
    json.dump(dict(note=note, passes=passes), open(DST + "pmc_mfma.json", "w"), indent=1)
