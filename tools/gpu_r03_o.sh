cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/bench_refit_forms.py f32 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['dtype'], d['batch'], d['N'], ' '.join('%s %.4f' % (k[3:], d[k]) for k in d if k.startswith('ms_')))"
