cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/time_refit_one.py 2>/dev/null | tail -2
python tools/bench_speed_test.py --quick 2>/dev/null | cut -c1-260 | tail -4
python tools/prof_fit.py 2>/dev/null | tail -6
