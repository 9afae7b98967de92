# The learning-loop and C5-append part of tools/run_profiles.sh alone (what changes when only the online path's kernels change).
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python tools/bench_learning_loop.py --data synthetic --schedule reference --parts 4 2>/dev/null > $O/learn_reference_parts4.json
python tools/bench_learning_loop.py --data synthetic --schedule reference 2>/dev/null > $O/learn_reference.json
python tools/bench_learning_loop.py --data synthetic --schedule online 2>/dev/null > $O/learn_online.json
python tools/bench_learning_loop.py --data synthetic --schedule online_tail 2>/dev/null > $O/learn_online_tail.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learn_reference -- python3 tools/bench_learning_loop.py --data synthetic --schedule reference --steps 80 --warmup 40 > $O/learn_reference_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learn_online -- python3 tools/bench_learning_loop.py --data synthetic --schedule online --steps 80 --warmup 40 > $O/learn_online_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learn_online_tail -- python3 tools/bench_learning_loop.py --data synthetic --schedule online_tail --steps 80 --warmup 40 > $O/learn_online_tail_prof.json 2>/dev/null
# the loop that learns from itself (rows built on the device from its own x_t, u_t, x_t+1; host-free staggered refits)
for dt in f32 f64; do
python tools/bench_learning_loop.py --schedule reference --dtype $dt 2>/dev/null > $O/learn_loop_reference_$dt.json
python tools/bench_learning_loop.py --schedule reference --no-stagger --dtype $dt 2>/dev/null > $O/learn_loop_reference_nostagger_$dt.json
python tools/bench_learning_loop.py --schedule online_tail --dtype $dt 2>/dev/null > $O/learn_loop_online_tail_$dt.json
done
python tools/bench_learning_loop.py --schedule reference --dtype f32 --fit-iters 100 --steps 40 --warmup 560 2>/dev/null > $O/learn_loop_reference_fit100_f32.json
python tools/bench_learning_loop.py --schedule reference --dtype f32 --batch 256 --fit-iters 100 --steps 40 --warmup 560 2>/dev/null > $O/learn_loop_reference_fit100_f32_b256.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learn_loop_reference -- python3 tools/bench_learning_loop.py --schedule reference --steps 80 > $O/learn_loop_reference_prof.json 2>/dev/null
bash tools/run_pmc_append_traffic.sh $R > /dev/null 2>&1
python tools/bench_online.py --repeat 3 2>/dev/null > $O/online_growth_f64.json
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_kernel_trace.csv" -size +2M -delete 2>/dev/null
for f in learn_reference_parts4 learn_reference learn_online learn_online_tail; do python3 -c "
import json,sys
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); s=d['shares']
print('$f', round(d['value']), 'ms/step %.4f pass %.4f solve %.4f refit/step %.4f' % (d['ms_per_step'], s['pass_ms_per_step'], s['solve_ms_per_step'], s['refit_ms_per_step']), 'final', d['final_vs_fp64_refit_on_device'])"; done
tail -3 $O/pmc_traffic_append.txt
