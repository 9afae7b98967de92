cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "refit or fit or golden or config or parity" 2>&1 | tail -2
timeout 300 python tools/check_refit_forms.py 2>&1 | tail -1
python tools/time_refit_wave.py "v=" 2>/dev/null | tail -1
python tools/time_refit_wave.py "v=" 2>/dev/null | tail -1
