cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in psrA psrB psrA psrB; do
  echo $v; BCBF_LIB_PATH=tools/_variants/libbcbf_$v.so python tools/time_shared.py 2>/dev/null | grep "N 512"
done
