#!/usr/bin/env python3
"""The learning closed loop at BASELINE configs[2] scale (the reference's real workload,
unicycle_move_to_pose.py:340-386): `--batch` independent control loops PER GPU, each with its own GP over its most recent
observations (at most `--max-train`); every step = one control step + one new observation per instance, every
`--refit-every`-th step the window is refactored.  `--schedule online_tail` (default: the posterior query and the new observation's
column share one pass over the window's factor, the points since the last window refit are contiguous rows beside it:
`bcbf_gp_tail_step`), `online` (the same with in-place appends into the operator) or `reference` (static GP between refits, the headline control step).

`--data loop` (default): the loop learns from ITSELF -- every step's observation row is built on the device from the loop's own
(x_t, u_t, x_{t+1}) inside the solve / plant launch, as LearnedShiftInvariantDynamics.train / fit builds it
(rollouts.self_learning_closed_loop: part batches on their own streams, host-free refits, staggered); schedules `reference` and
`online_tail`.  `--fit-iters K` (reference schedule): every refit first runs K Adam iterations of the marginal likelihood for every
instance (the reference's `fit(..., training_iter=100)`, BatchedHyperFit).  `--data synthetic`: pre-drawn well-conditioned rows
(rollouts.learning_closed_loop; rounds 4-5).

    python tools/bench_learning_loop.py                          # 4096 x 512, fp32, 200 timed steps, refit every 40
    python tools/bench_learning_loop.py --gpus 8                 # starts its 8 ranks itself (weak scaling, no collective in the loop)
    python bench.py --config learn [same flags]

Rank 0 prints ONE JSON line: instance-steps/s WITH learning over all ranks (slowest rank's time), the shares of the pass /
solve / refit, a roofline entry per kernel, and the self-check of the final model against a from-scratch fp64 refit of the
final window on the device (the oracle parity of the same loop: tests/test_gpu_configs.py)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
ap.add_argument("--max-train", type=int, default=512, help="most points a model ever holds (the reference's max_train)")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=40)
ap.add_argument("--refit-every", type=int, default=40)
ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
ap.add_argument("--schedule", choices=["online", "online_tail", "reference"], default="online_tail")
ap.add_argument("--parts", type=int, default=0, help="part batches on their own streams (0 = default: 4 with --data loop; synthetic data: 1)")
ap.add_argument("--data", choices=["loop", "synthetic"], default="loop")
ap.add_argument("--fit-iters", type=int, default=0, help="Adam iterations of the marginal likelihood per refit (--data loop, reference schedule)")
ap.add_argument("--no-stagger", action="store_true", help="--data loop: every part batch refits at the same step")
ap.add_argument("--raw-inputs", action="store_true", help="--data loop: regressor inputs = the raw state, not the shift-invariant (0, 0, theta)")
ap.add_argument("--dt", type=float, default=0.01)
ap.add_argument("--retry-levels", type=int, default=3)
ap.add_argument("--min-jitter-level", type=float, default=1e-5, help="--data loop: floor of the per-instance jitter level (make_psd starts at 1e-5)")
ap.add_argument("--factor-f64", action="store_true", help="--data loop --dtype f32: factor the windows in fp64, round the operator to fp32 for the passes")
a = ap.parse_args()
if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or os.environ.get("BCBF_BENCH_FORCE_LAUNCH") == "1"):
    from bayesian_cbf_amd.distributed import launch_ranks
    sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
from bayesian_cbf_amd.distributed import RankContext
from bayesian_cbf_amd.rollouts import (learning_closed_loop, final_window_vs_device_refit, self_learning_closed_loop,
                                       final_model_vs_fp64_refit)
ctx = RankContext()
dtype = torch.float32 if a.dtype == "f32" else torch.float64
if a.data == "loop":
    if a.schedule == "online":
        raise SystemExit("--data loop: schedules reference | online_tail")
    out, final = self_learning_closed_loop(a.batch, a.max_train, a.steps, a.refit_every, warmup=None if a.warmup == 40 else a.warmup,
                                           dtype=dtype, device=ctx.device, seed=1234 + ctx.rank, schedule=a.schedule, parts=a.parts or 4,
                                           stagger=not a.no_stagger, shift_invariant=not a.raw_inputs, dt=a.dt, retry_levels=a.retry_levels,
                                           fit_iters=a.fit_iters, barrier=ctx.barrier,
                                           factor_dtype=torch.float64 if (a.factor_f64 and a.dtype == "f32") else None, min_jitter_level=a.min_jitter_level)
    chk = final_model_vs_fp64_refit(final)
    nfail = out["refit_failures_after_retries"]
    what = "rows built on the device from the loop's own (x_t, u_t, x_t+1)"
else:
    out, final = learning_closed_loop(a.batch, a.max_train, a.steps, a.refit_every, warmup=a.warmup, dtype=dtype, device=ctx.device,
                                      seed=1234 + ctx.rank, schedule=a.schedule, barrier=ctx.barrier, parts=a.parts or 1)
    chk = final_window_vs_device_refit(final)
    nfail = out["append_or_refit_failures"]
    what = "pre-drawn synthetic rows"
el, per_rank = ctx.reduce_times(out["seconds"])
fails = ctx.reduce_sum([nfail])
if ctx.rank == 0:
    total = a.batch * ctx.world
    out.update(metric="control steps/sec WITH learning (GP posterior + %s + CBF-QP, window refit every %d steps%s) at "
                      "N_train<=%d, batch=%d; observations: %s"
                      % ("append" if a.schedule != "reference" else "static model", a.refit_every,
                         ", %d Adam iterations of the marginal likelihood per refit" % a.fit_iters if a.fit_iters else "", a.max_train, a.batch, what),
               value=total * a.steps / el, unit="control steps/s (instance-steps)", n_gpus=ctx.world, ms_per_step=el / a.steps * 1e3,
               seconds=el, instance_steps_per_s=total * a.steps / el, higher_is_better=True, scaling="weak",
               data="synthetic task; " + what, comm=ctx.comm_info(per_rank), append_or_refit_failures=int(fails[0]),
               final_vs_fp64_refit_on_device=chk)
    print(json.dumps(out), flush=True)
ctx.close()
