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
    ap.add_argument("--dtype", choices=["f32", "f64"], default=None,
                    help="precision of the shared learned model (default: f64 as the reference's module for N <= 512, else f32)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
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
    if rank == 0:
        print(json.dumps(dict(config="c4: Monte-Carlo safety rollouts (unicycle_bayes_cbf_safe_obstacle recipe)",
                              trajectories=args.trajectories, steps=args.steps, n_gpus=world, seconds=el, loop_seconds=loop_max,
                              trajectory_steps_per_s=args.trajectories * args.steps / el,
                              trajectory_steps_per_s_loop_only=args.trajectories * args.steps / loop_max, scaling="strong",
                              comm=ctx.comm_info(per_rank), **out["stats"])))
    ctx.close()


if __name__ == "__main__":
    main()
