#!/usr/bin/env python3
"""BASELINE configs[4]: online GP growth N0 -> N1 (one observation per instance per control step, entered in place into
capacity-reserving storage; one posterior + SOCP control step per observation), `--batch` instances PER GPU (weak
scaling: instances never interact, every rank grows its own batch, no collective inside the loop).

    python tools/bench_online.py                      # one GPU
    python tools/bench_online.py --gpus 8             # starts its 8 ranks itself (or run under torch.distributed.run)

Rank 0 prints ONE JSON line per run: per-N-range timings (the slowest rank's), appends/s over all ranks, n_gpus, the
communicator's backend / world size, every rank's own seconds, and the end-to-end check against a from-scratch refit of
the final data set (worst rank)."""
import os, sys, json, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--batch", type=int, default=256, help="instances per GPU")
ap.add_argument("--n0", type=int, default=128)
ap.add_argument("--n1", type=int, default=2048)
ap.add_argument("--dtype", choices=["f32", "f64"], default="f64")
ap.add_argument("--no-control", action="store_true")
ap.add_argument("--packed", action="store_true", help="the packed layout of exactly N points (copy per append) instead of reserved storage")
ap.add_argument("--unfused", action="store_true", help="reserved storage, but the control query and the append each make their own pass")
ap.add_argument("--repeat", type=int, default=1, help="runs (reproducibility of the per-segment figures)")
ap.add_argument("--window", type=int, default=0, help="sliding window of the most recent W points (ops.ReservedGP(window=W)): grow "
                                                      "to W, then drop the oldest 32 every 32 appends; --n1 = observations seen")
ap.add_argument("--tail", action="store_true", help="growth with a row-major tail committed 32 rows at a time (bcbf_gp_tail_step + bcbf_gp_tail_commit) "
                                                    "instead of in-place element-per-line appends")
a = ap.parse_args()
# BCBF_BENCH_FORCE_LAUNCH=1 (test hook, as in bench.py): also a one-GPU run goes through the launcher parent -> child rank path
if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or os.environ.get("BCBF_BENCH_FORCE_LAUNCH") == "1"):
    from bayesian_cbf_amd.distributed import launch_ranks
    sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
from bayesian_cbf_amd.distributed import RankContext
from bayesian_cbf_amd.rollouts import online_gp_growth
ctx = RankContext()
for _ in range(a.repeat):
    ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = online_gp_growth(a.batch, a.n0, a.n1, dtype=torch.float64 if a.dtype == "f64" else torch.float32, device=ctx.device,
                           seed=5 + ctx.rank, with_control=not a.no_control, reserved=not a.packed, fused=not a.unfused,
                           window=a.window or None, tail=a.tail)
    torch.cuda.synchronize()
    el, per_rank = ctx.reduce_times(time.perf_counter() - t0)
    # the slowest rank's figure per segment, the worst rank's deviation, the sum of failures: one short reduction each
    segs = out["segments"]
    import torch.distributed as dist
    v = torch.tensor([[s["append_ms"], s["control_step_ms"]] for s in segs] + [[out["final_vs_refit"]["Mk"], out["final_vs_refit"]["Bk"]]],
                     dtype=torch.float64, device=ctx.device if ctx.backend == "nccl" else "cpu")
    if ctx.multi:
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
    fails = ctx.reduce_sum([out["append_failures"], out["refit_failures"]])
    if ctx.rank == 0:
        for s, (ta, tc) in zip(segs, v[:-1].tolist()):
            s["append_ms"], s["control_step_ms"], s["step_ms"] = ta, tc, ta + tc      # all three the slowest rank's
            s.pop("append_GBs_algorithmic", None)
            rf = s.get("roofline")
            if rf and rf.get("achieved") is not None:          # over the slowest rank's time, like append_ms
                rf["achieved"] = rf["algorithmic_bytes_per_launch"] / (ta * 1e-3) / 1e9
                rf["frac"] = rf["achieved"] / rf["peak"]
        appends = (a.n1 - a.n0) * a.batch * ctx.world
        out.update(config="c5: online GP growth", n_gpus=ctx.world, batch_per_gpu=a.batch, scaling="weak", seconds=el,
                   instance_appends_per_s=appends / el, comm=ctx.comm_info(per_rank), append_failures=int(fails[0]),
                   refit_failures=int(fails[1]), final_vs_refit=dict(Mk=float(v[-1, 0]), Bk=float(v[-1, 1])))
        print(json.dumps(out), flush=True)
ctx.close()
