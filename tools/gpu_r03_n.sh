cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export BCBF_REFIT_PAIR=0
for v in rwbase rwtrim8 rwtrim4 rwtrim2 rwbase rwtrim4; do
  BCBF_LIB_PATH=tools/_variants/libbcbf_$v.so python tools/time_refit32.py $v 2>&1 | tail -1
done
