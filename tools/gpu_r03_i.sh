cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt
python examples_mc_rollouts.py --trajectories 32768 --steps 200 --graph 2>/dev/null | tail -1 > $O/mc_graph.json
python examples_mc_rollouts.py --trajectories 32768 --steps 200 2>/dev/null | tail -1 > $O/mc_eager.json
python examples_mc_rollouts.py --trajectories 32768 --steps 200 --graph 2>/dev/null | tail -1 >> $O/mc_graph.json
cat $O/pytest.txt
python - <<'PY'
import json
for f in ("mc_graph","mc_eager"):
    for l in open("gpurun_out/r03i/%s.json"%f):
        d=json.loads(l); print(f, round(d["loop_seconds"]*1e3,2), "ms", round(d["trajectory_steps_per_s_loop_only"]/1e6,1), "M/s", d["collisions"], d["solver_failures"], round(d["min_h"],4), round(d["mean_cost"],5))
PY
