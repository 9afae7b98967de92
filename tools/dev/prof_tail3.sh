cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tailprof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tailprof -- python3 tools/dev/micro_tail.py > gpurun_out/tailprof/out.txt 2>gpurun_out/tailprof/err.log
cat gpurun_out/tailprof/out.txt
python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/tailprof/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("tail :", [round(d(r)) for r in rows if "gp_tail_step_kernel" in r["Kernel_Name"]])
print("fused:", [round(d(r)) for r in rows if "posterior_step_kernel" in r["Kernel_Name"] and ", false, 1" in r["Kernel_Name"].split("(")[0]])
PY
find gpurun_out/tailprof -name "*.db" -delete; find gpurun_out/tailprof -name "*kernel_trace.csv" -delete
