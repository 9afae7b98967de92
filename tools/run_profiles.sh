# One GPU call that produces everything tools/collect_profiles.py copies into profiles/ (ROUND tag = $1, default r03).
# Counters are collected in their own passes (--pmc + --kernel-trace only), as MI355X_MICROARCH.md prescribes.
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # the setting bench.py gives itself; under rocprofv3 the runtime is up before Python runs
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_driver_form.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 bench.py --steps 100 --warmup 60 --cpu-sample 0 > $O/bench_default_prof.json 2> $O/bench_default_prof.err
python tools/trace_union.py $O/prof_default $O/bench_default_prof.json $O/bench_default_prof_union.json > $O/union_default.txt 2>&1
python bench.py --parts 1 --cpu-sample 0 > $O/bench_parts1.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_parts1 -- python3 bench.py --parts 1 --steps 100 --warmup 60 --cpu-sample 0 > $O/bench_parts1_prof.json 2> $O/bench_parts1_prof.err
python tools/trace_union.py $O/prof_parts1 $O/bench_parts1_prof.json $O/bench_parts1_prof_union.json > $O/union_parts1.txt 2>&1
for P in 2 3; do python bench.py --parts $P --cpu-sample 0 > $O/bench_parts$P.json 2>/dev/null; done
python bench.py --regime shared --cpu-sample 0 > $O/bench_shared.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shared -- python3 bench.py --regime shared --steps 100 --warmup 60 --cpu-sample 0 > $O/bench_shared_prof.json 2> $O/bench_shared_prof.err
python bench.py --regime shared --dtype f64 --cpu-sample 0 > $O/bench_shared_f64.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shared_f64 -- python3 bench.py --regime shared --dtype f64 --steps 100 --warmup 60 --cpu-sample 0 > $O/bench_shared_f64_prof.json 2> $O/bench_shared_f64_prof.err
python bench.py --dtype f64 --cpu-sample 0 > $O/bench_f64.json 2>/dev/null
# regime S at a batch whose part batches (two of 5120 loops) take the five-queries-per-wave form
python bench.py --regime shared --batch 10240 --cpu-sample 0 > $O/bench_shared_b10240.json 2>/dev/null
python bench.py --regime shared --batch 10240 --dtype f64 --cpu-sample 0 > $O/bench_shared_b10240_f64.json 2>/dev/null
# HBM traffic of the roofline kernel (default schedule AND the one-stream schedule): three passes each
for sched in "" "--parts 1"; do
  tag=$(echo "default$sched" | tr -d ' -')
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d=$O/pmc_traffic_${tag}_$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 $sched > $d.log 2>&1
  done
done
# ... and of the jets instantiations (rel-degree-2 path): three passes of tools/bench_reldeg2.py
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/pmc_traffic_jets_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/bench_reldeg2.py > $d.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_reldeg2 -- python3 tools/bench_reldeg2.py > $O/reldeg2_prof.jsonl 2> $O/reldeg2_prof.err
# MFMA utilisation of the matrix-core kernels
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc_shared -- python3 bench.py --regime shared --parts 1 --steps 10 --warmup 2 --cpu-sample 0 > $O/pmc_shared.log 2>&1
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc_shared_f64 -- python3 bench.py --regime shared --dtype f64 --parts 1 --steps 10 --warmup 2 --cpu-sample 0 > $O/pmc_shared_f64.log 2>&1
for cfg in C2 C3f64 C3 N1024f64; do
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc_refit_$cfg -- python3 tools/bench_configs.py $cfg > $O/pmc_refit_$cfg.log 2>&1
done
python tools/bench_configs.py 2>/dev/null > $O/configs.jsonl
python tools/time_shared.py 2>/dev/null > $O/shared_queries.txt
python tools/time_shared_sweep.py 2>/dev/null > $O/shared_sweep.jsonl
python tools/bench_refit_forms.py 2>/dev/null > $O/refit_forms.jsonl
python tools/bench_refit_forms.py f32 2>/dev/null > $O/refit_forms_f32.jsonl
python tools/bench_online.py --repeat 2 2>/dev/null > $O/online_growth_f64.json
python tools/bench_online.py --unfused --repeat 3 2>/dev/null > $O/online_growth_f64_unfused3.json
python tools/bench_online.py --packed 2>/dev/null > $O/online_growth_f64_packed.json
python tools/bench_online.py --batch 1024 2>/dev/null > $O/online_growth_f64_batch1024.json
BCBF_APPEND_PAIR=1 python tools/bench_online.py 2>/dev/null > $O/online_growth_f64_pairform.json
python tools/bench_online.py --n1 1536 --window 512 2>/dev/null > $O/online_window512_f64.json
python tools/bench_online.py --n0 512 --n1 2560 --window 1024 2>/dev/null > $O/online_window1024_f64.json
python tools/prof_speed_host.py 2>/dev/null | head -3 > $O/speed_call_host.txt
python tools/bench_reldeg2.py 2>/dev/null > $O/reldeg2.jsonl
python tools/bench_speed_test.py --quick 2>/dev/null > $O/speed_test.jsonl
python tools/bench_speed_test_unicycle.py --quick 2>/dev/null > $O/speed_test_unicycle.jsonl
python tools/learn_dynamics_matrix_vector.py /tmp/learn_matrix_vector > /dev/null 2>&1; cp gpurun_out/learn_matrix_vector.jsonl $O/ 2>/dev/null
python examples_mc_rollouts.py --trajectories 32768 --graph --predict-8gpu 2>/dev/null | tail -1 > $O/mc_rollouts.txt
for T in 4096 8192 16384; do python examples_mc_rollouts.py --trajectories $T --graph 2>/dev/null | tail -1 >> $O/mc_rollouts.txt; done
python examples_mc_rollouts.py --trajectories 32768 2>/dev/null | tail -1 >> $O/mc_rollouts.txt
bash tools/run_pmc_refit_traffic.sh $R > /dev/null 2>&1      # refit traffic past L2 -> $O/pmc_traffic_refit.json
# the learning closed loop at C3 scale (round 5): reference cadence on four part batches, on one stream, and the online schedule
python tools/bench_learning_loop.py --data synthetic --schedule reference --parts 4 2>/dev/null > $O/learn_reference_parts4.json
python tools/bench_learning_loop.py --data synthetic --schedule reference 2>/dev/null > $O/learn_reference.json
python tools/bench_learning_loop.py --data synthetic --schedule online 2>/dev/null > $O/learn_online.json
python tools/bench_learning_loop.py --data synthetic --schedule online_tail 2>/dev/null > $O/learn_online_tail.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learn_reference -- python3 tools/bench_learning_loop.py --data synthetic --schedule reference --steps 80 --warmup 40 > $O/learn_reference_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learn_online -- python3 tools/bench_learning_loop.py --data synthetic --schedule online --steps 80 --warmup 40 > $O/learn_online_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learn_online_tail -- python3 tools/bench_learning_loop.py --data synthetic --schedule online_tail --steps 80 --warmup 40 > $O/learn_online_tail_prof.json 2>/dev/null
# round 6: the loop that learns from ITSELF (rows built on the device from its own x_t, u_t, x_t+1; host-free staggered refits)
for dt in f32 f64; do
python tools/bench_learning_loop.py --schedule reference --dtype $dt 2>/dev/null | tail -1 > $O/learn_loop_reference_$dt.json
python tools/bench_learning_loop.py --schedule reference --no-stagger --dtype $dt 2>/dev/null | tail -1 > $O/learn_loop_reference_nostagger_$dt.json
python tools/bench_learning_loop.py --schedule online_tail --dtype $dt 2>/dev/null | tail -1 > $O/learn_loop_online_tail_$dt.json
done
# fp32 passes on fp64 factors (jitter floor 1e-3), next to pure fp32 / fp64 at the same floor
python tools/bench_learning_loop.py --schedule reference --dtype f32 --factor-f64 --min-jitter-level 1e-3 2>/dev/null | tail -1 > $O/learn_loop_reference_mixed.json
python tools/bench_learning_loop.py --schedule reference --dtype f32 --min-jitter-level 1e-3 2>/dev/null | tail -1 > $O/learn_loop_reference_f32_floor1e-3.json
python tools/bench_learning_loop.py --schedule reference --dtype f64 --min-jitter-level 1e-3 2>/dev/null | tail -1 > $O/learn_loop_reference_f64_floor1e-3.json
python tools/bench_learning_loop.py --schedule reference --dtype f32 --fit-iters 100 --steps 40 2>/dev/null | tail -1 > $O/learn_loop_reference_fit100_f32.json
python tools/bench_learning_loop.py --schedule reference --dtype f32 --batch 256 --fit-iters 100 --steps 40 2>/dev/null | tail -1 > $O/learn_loop_reference_fit100_f32_b256.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learn_loop_reference -- python3 tools/bench_learning_loop.py --schedule reference --steps 80 > $O/learn_loop_reference_prof.json 2>/dev/null
# round 6: one batched Adam iteration of the marginal likelihood (BatchedHyperFit), per-kernel shares
python tools/dev/try_fit_scale.py 2>/dev/null > $O/fit_iteration.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fit_iter -- python3 tools/dev/try_fit_scale.py > /dev/null 2>&1
python tools/dev/sweep_refit_footprint.py 2>/dev/null > $O/refit_footprint.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_speed_call -- python3 tools/prof_speed_host.py > /dev/null 2>&1
# round 6: C5 growth with a row-major tail committed 32 rows at a time, next to the in-place form
( for args in "--dtype f32 --batch 4096 --n0 256 --n1 512" "--dtype f64 --batch 1024 --n0 1024 --n1 1280" "--dtype f64"; do
    python tools/bench_online.py $args 2>/dev/null | tail -1; python tools/bench_online.py --tail $args 2>/dev/null | tail -1; done ) > $O/online_growth_tail.jsonl
# C5 append: counters + kernel trace (round 5) -> $O/pmc_traffic_append.json, online_b*_n*.json
bash tools/run_pmc_append_traffic.sh $R > /dev/null 2>&1
# fp32 online growth at the C3 batch
python tools/bench_online.py --dtype f32 --batch 4096 --n0 256 --n1 512 2>/dev/null | tail -1 > $O/online_f32_b4096.json
# measured deviation behind every asserted tolerance of the GPU suite
BCBF_TOL_REPORT=$PWD/$O/tol.jsonl python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; python tools/tol_report.py $O/tol.jsonl 0 > $O/tol_report.txt 2>/dev/null
du -sh $O; ls $O | head -50
# keep the merge small: the raw traces are not needed, only the csv summaries
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_kernel_trace.csv" -size +2M -delete 2>/dev/null
du -sh $O
