# development: the main lines of every streaming form on the current library
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('headline %.3f M  ms %.4f frac %.4f kernel_ms %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms']))"
python bench.py --cpu-sample 0 --parts 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('headline parts1 %.3f M  ms %.4f frac %.4f kernel_ms %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms']))"
python bench.py --cpu-sample 0 --dtype f64 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('headline f64 %.3f M  ms %.4f frac %.4f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac']))"
timeout 150 python tools/bench_reldeg2.py 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('jets', d['N'], d['n'], d['m'], d['dtype'], 'jets_ms %.4f GB/s %.0f values_ms %.4f' % (d['jets_ms'], d['jets_GBs_algorithmic'], d['values_only_ms']))"
for cfg in "256 1024 2048" "1024 1024 1280"; do set -- $cfg
timeout 200 python tools/bench_online.py --batch $1 --n0 $2 --n1 $3 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['segments'][0]
print('C5 b$1 append_ms %.4f step_ms %.4f frac %.3f' % (s['append_ms'], s['step_ms'], s['roofline']['frac']))"; done
for args in "--schedule online_tail" "--schedule online_tail --parts 2" "--schedule reference --parts 4" "--schedule online"; do
timeout 200 python tools/bench_learning_loop.py --steps 200 --warmup 40 $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['shares']
print('learn $args: pass_ms %.4f ms_per_step %.4f value %.3f M/s' % (s['pass_ms_per_step'], d['ms_per_step'], d['value'] / 1e6))"; done
timeout 300 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l)
    print(d.get('config'), {k: round(v, 4) for k, v in d.items() if k.endswith('_ms')}, {k: round(v.get('frac', 0) or 0, 3) for k, v in d.get('roofline', {}).items() if isinstance(v, dict)})"
