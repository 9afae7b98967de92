cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python tools/dev/try_fit_scale.py 2>/dev/null > $O/fit_iteration.jsonl
rm -rf $O/prof_fit_iter
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fit_iter -- python3 tools/dev/try_fit_scale.py > /dev/null 2>&1
python tools/dev/time_trtri.py 2>/dev/null > $O/trtri_syrk.jsonl
python tools/bench_learning_loop.py --schedule reference --dtype f32 --fit-iters 100 --steps 40 2>/dev/null | tail -1 > $O/learn_loop_reference_fit100_f32.json
python tools/bench_learning_loop.py --schedule reference --dtype f32 --batch 256 --fit-iters 100 --steps 40 2>/dev/null | tail -1 > $O/learn_loop_reference_fit100_f32_b256.json
python tools/bench_learning_loop.py --schedule reference --dtype f32 --factor-f64 --min-jitter-level 1e-3 --fit-iters 100 --steps 40 --batch 256 2>/dev/null | tail -1 > $O/learn_loop_reference_fit100_mixed_b256.json
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_kernel_trace.csv" -size +2M -delete 2>/dev/null
cat $O/fit_iteration.jsonl | cut -c1-110
