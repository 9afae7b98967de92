cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "refit or fit or golden or config" 2>&1 | tail -3
python tools/bench_configs.py C2 2>/dev/null | cut -c1-400
python tools/bench_refit_forms.py 2>/dev/null | head -12
python tools/bench_refit_forms.py f32 2>/dev/null | head -12
