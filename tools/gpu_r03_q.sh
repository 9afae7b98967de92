cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export BCBF_LIB_PATH=tools/_variants/libbcbf_rptrace.so
python tools/trace_refit_pair.py f64 64 256 2>&1 | tail -22
python tools/trace_refit_pair.py f64 1024 256 2>&1 | tail -22
