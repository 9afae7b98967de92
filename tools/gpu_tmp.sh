cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c4prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4prof -- python3 examples_mc_rollouts.py --trajectories 32768 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c4prof/**/*_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
