#!/usr/bin/env python3
"""BASELINE config 4: Monte-Carlo safety rollouts of the unicycle Bayes-CBF controller, trajectories sharded
over the GPUs of one node (contiguous shard per rank, no collective inside the loop), one RCCL reduction at the end.

    python examples_mc_rollouts.py --trajectories 4096 --steps 200
    python examples_mc_rollouts.py --gpus 8 --trajectories 32768 --steps 200 --graph        # starts its 8 ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        examples_mc_rollouts.py --gpus 8 --trajectories 32768 --steps 200

Rank 0 prints ONE JSON line: whole-job trajectory-steps/s over the slowest rank's time, n_gpus, the communicator's
backend / world size and every rank's own seconds.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--trajectories", type=int, default=4096, help="over ALL ranks (strong split of one Monte-Carlo job)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--max-risk", type=float, default=0.01)
    ap.add_argument("--learned", type=int, default=0, help="N_train of a per-trajectory learned GP (0 = fixed kernel)")
    ap.add_argument("--graph", action="store_true", help="replay the closed-loop step from a captured HIP graph")
    ap.add_argument("--shared-learned", type=int, default=0,
                    help="N_train of ONE learned GP queried by every trajectory (matrix-core posterior; N <= 512 in fp64)")
    ap.add_argument("--predict-8gpu", action="store_true",
                    help="one GPU only: also time the per-GPU shape of the 8-way strong split (trajectories / 8) and print "
                         "predicted_strong_scaling_8gpu = loop(trajectories) / loop(trajectories / 8) -- what an 8-GPU node could gain")
    ap.add_argument("--dtype", choices=["f32", "f64"], default=None,
                    help="precision of the shared learned model (default: f64 as the reference's module for N <= 512, else f32)")
    args = ap.parse_args()
    # BCBF_BENCH_FORCE_LAUNCH=1 (test hook, as in bench.py): also a one-GPU run goes through the launcher parent -> child rank path
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("BCBF_BENCH_FORCE_LAUNCH") == "1"):
        from bayesian_cbf_amd.distributed import launch_ranks
        sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    from bayesian_cbf_amd.distributed import RankContext, shard_range
    ctx = RankContext()
    world, rank = ctx.world, ctx.rank
    from bayesian_cbf_amd.rollouts import monte_carlo_safety_rollouts
    a, b = shard_range(args.trajectories, rank, world)
    gp = None
    if args.learned:
        from bayesian_cbf_amd.control_affine_model import BatchedControlAffineGP
        from bayesian_cbf_amd.synthetic import make_instances
        p = make_instances(b - a, args.learned, 3, 2, dtype=torch.float64, device=ctx.device, seed=100 + rank)
        gp = BatchedControlAffineGP(p["X"], p["U"], 0.05 * p["Xdot"], 1e-2 * p["A"], 1e-2 * p["Bm"], p["ell"], p["s2"],
                                    p["M0"]).as_dict()
    dtype = torch.float64
    if args.shared_learned:
        from bayesian_cbf_amd.control_affine_model import BatchedControlAffineGP
        from bayesian_cbf_amd.synthetic import make_instances
        f64 = args.dtype == "f64" or (args.dtype is None and args.shared_learned <= 512)
        dtype = torch.float64 if f64 else torch.float32
        p = make_instances(1, args.shared_learned, 3, 2, dtype=dtype, device=ctx.device, seed=100)     # same model on every rank
        gp = BatchedControlAffineGP(p["X"], p["U"], 0.05 * p["Xdot"], 1e-2 * p["A"], 1e-2 * p["Bm"], p["ell"], p["s2"],
                                    p["M0"]).as_dict()
    torch.cuda.synchronize()
    ctx.barrier()
    t0 = time.perf_counter()
    out = monte_carlo_safety_rollouts(b - a, numSteps=args.steps, gp=gp, max_risk=args.max_risk, seed=rank, dtype=dtype,
                                      device=ctx.device, use_graph=args.graph)      # (its statistics: the one reduction)
    torch.cuda.synchronize()
    el_own = time.perf_counter() - t0
    el, per_rank = ctx.reduce_times(el_own)
    loop_max, _ = ctx.reduce_times(out["loop_seconds"])
    # roofline of the loop's dominant kernel, over the LOOP time (every launch of a step included: a lower bound on the
    # kernel's own rate).  Shared learned model: the regime-S posterior on the matrix cores, (1+m) N^2 flop per trajectory-
    # step.  Fixed-kernel recipe (the reference's unicycle_bayes_cbf_safe_obstacle): the fused task rows / terms / SOCP /
    # plant-step kernel holds everything in registers -- neither HBM nor the matrix cores bind it (a chain of fp64 divisions
    # and square roots per interior-point iteration); its algorithmic bytes are the per-trajectory inputs and outputs
    n_loc, isz = b - a, (8 if dtype == torch.float64 else 4)
    if args.shared_learned:
        N_ = args.shared_learned
        flops = float(n_loc) * 3 * N_ * N_
        ach = flops * args.steps / out["loop_seconds"] / 1e12
        peak = 78.6 if isz == 8 else 157.3
        roof = dict(bound="mfma", kernel="posterior_shared_reg_kernel<%s>" % ("double" if isz == 8 else "float"),
                    algorithmic_flops_per_launch=flops, achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, traffic=None,
                    how="over the loop time of rank 0 (posterior + solve + bookkeeping launches per step)")
    else:
        per_traj = isz * (3 + 3 + 3 + 9 + 9 + 9 + 4 + 2 + 3 + 2 + 1 + 3 + 3 + 4 + 3) + 4 * 3
        if args.learned:
            per_traj += isz * (args.learned * (args.learned + 1) // 2 + args.learned * 9)
        byt = float(n_loc) * per_traj
        ach = byt * args.steps / out["loop_seconds"] / 1e9
        roof = dict(bound="hbm", kernel=("posterior_step_kernel<double, 3, ...>" if args.learned else "socp_quad_kernel<double, 2, true> (fused task rows + terms + SOCP + plant step)"),
                    algorithmic_bytes_per_launch=byt, achieved=ach, peak=8000.0, unit="GB/s", frac=ach / 8000.0, traffic=None,
                    limiter=(None if args.learned else "latency: register-resident interior-point iterations (fp64 division / sqrt chains); "
                             "neither the HBM nor the MFMA roofline binds this kernel -- see solver iterations in DESIGN.md 3.2"),
                    how="over the loop time of rank 0 (every launch of a step)")
    extra = dict(setup_and_capture_seconds=el - loop_max, setup_share_of_wall=(el - loop_max) / el if el > 0 else None)
    if args.predict_8gpu and world == 1:
        # the per-GPU shape of the strong split on THIS GPU: an eighth of the trajectories (same recipe, same graph option)
        n8 = max(1, args.trajectories // 8)
        gp8 = gp if (gp is None or args.shared_learned) else {k: (v[:n8].contiguous() if torch.is_tensor(v) and v.shape[0] == n_loc else v) for k, v in gp.items()}
        torch.cuda.synchronize()
        out8 = monte_carlo_safety_rollouts(n8, numSteps=args.steps, gp=gp8, max_risk=args.max_risk, seed=rank, dtype=dtype,
                                           device=ctx.device, use_graph=args.graph)
        torch.cuda.synchronize()
        extra.update(loop_seconds_at_one_eighth=out8["loop_seconds"], trajectories_at_one_eighth=n8,
                     us_per_step_full=out["loop_seconds"] / args.steps * 1e6, us_per_step_at_one_eighth=out8["loop_seconds"] / args.steps * 1e6,
                     predicted_strong_scaling_8gpu=out["loop_seconds"] / out8["loop_seconds"],
                     note="the loop is latency bound (four lanes per trajectory, one wave per CU at 4096 trajectories): an eighth of the "
                          "trajectories takes nearly the same time per step, so 8 GPUs cannot give 8x on a FIXED 32768-trajectory job; "
                          "C4 is reported weak-scaled (trajectories per GPU fixed)")
    if rank == 0:
        print(json.dumps(dict(config="c4: Monte-Carlo safety rollouts (unicycle_bayes_cbf_safe_obstacle recipe)", **extra,
                              trajectories=args.trajectories, steps=args.steps, n_gpus=world, seconds=el, loop_seconds=loop_max,
                              trajectory_steps_per_s=args.trajectories * args.steps / el,
                              trajectory_steps_per_s_loop_only=args.trajectories * args.steps / loop_max, scaling="strong",
                              comm=ctx.comm_info(per_rank), roofline=roof, **out["stats"])))
    ctx.close()


if __name__ == "__main__":
    main()
