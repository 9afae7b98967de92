cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tailprof
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  d=gpurun_out/tailprof/$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/dev/micro_tail.py > $d.txt 2>$d.err
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("gpurun_out/tailprof/**/*counter_collection.csv", recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = "tail" if "gp_tail_step_kernel" in r["Kernel_Name"] else ("fused" if ("posterior_step_kernel" in r["Kernel_Name"] and ", false, 1" in r["Kernel_Name"].split("(")[0]) else None)
        if k is None: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-6s %-26s first3 %s   last3 %s" % (k, c, ["%.4g" % x for x in v[:3]], ["%.4g" % x for x in v[-3:]]))
PY
find gpurun_out/tailprof -name "*.db" -delete; find gpurun_out/tailprof -name "*kernel_trace.csv" -delete
