cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "refit or fit or golden or config or parity" 2>&1 | tail -2
timeout 300 python tools/check_refit_forms.py 2>&1 | tail -1
python tools/bench_configs.py C2 2>/dev/null | cut -c1-170
python tools/bench_refit_forms.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d['N'] <= 256: print(d['dtype'], d['batch'], d['N'], ' '.join('%s %.4f' % (k[3:], d[k]) for k in d if k.startswith('ms_')))"
python tools/bench_refit_forms.py f32 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d['N'] <= 512 and d['batch'] <= 1024: print(d['dtype'], d['batch'], d['N'], ' '.join('%s %.4f' % (k[3:], d[k]) for k in d if k.startswith('ms_')))"
