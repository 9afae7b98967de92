cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
python tools/prof_fit.py > $O/prof_fit.txt 2>&1
cat $O/pytest.txt; tail -12 $O/prof_fit.txt
