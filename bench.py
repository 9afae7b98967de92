#!/usr/bin/env python3
"""Headline benchmark: batched GP-posterior + CBF/CLF chance-constraint + SOCP control steps.

Workload = BASELINE.json configs[2]: unicycle (x in R^3, u in R^2), N_train = 512,
batch = 4096 independent control-loop instances per GPU (regime I: every instance owns its GP),
fp32.  One "step" = one pass of the hot path over the batch:
    unicycle_constraints -> posterior_step -> cbc_terms -> socp -> plant Euler step
with all inputs resident in HBM.  Multi-GPU = more instances (weak scaling), one process per GPU,
no collective inside the loop, one RCCL all-reduce of a small statistics vector at the end.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = instance control-steps/s over all GPUs
(= batch x batched-steps/s; `batched_steps_per_s` is reported beside it).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--ntrain", type=int, default=512)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--variant", default="dense")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="instances timed for the CPU baseline (0 = skip)")
    ap.add_argument("--chunks", type=int, default=1, help="independent sub-batches, one HIP stream each")
    ap.add_argument("--regime", choices=["independent", "shared"], default="independent",
                    help="independent: every instance owns its GP (headline, HBM bound); shared: one learned model, "
                         "`batch` closed loops (Monte-Carlo rollouts, BASELINE configs[3]; matrix-core bound)")
    return ap.parse_args()


def measured_traffic(N, Bt, dtype_name, bytes_launch):
    """HBM bytes per launch of the roofline kernel from the committed PMC passes (profiles/*_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 corrections applied).  A counter run cannot
    be nested inside this process, so the number is attached only when the profiled workload is this workload."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        w = d.get("workload", {})
        if w.get("N_train") == N and w.get("dtype") == dtype_name and w.get("batch"):
            best = d["hbm_bytes_per_launch"] * (Bt / float(w["batch"]))    # per-instance traffic is batch independent
    return best


def algorithmic_bytes_per_instance(N, n, m, itemsize):
    """packed factor + whitened targets + train inputs + UH*B, each read once."""
    return itemsize * (N * (N + 1) // 2 + N * n + N * n + N * (1 + m))


def cpu_baseline(p, task, sample, N, n, m):
    """The oracle (a numpy port of the reference's arithmetic, reference-style: one instance at a
    time, Cholesky cached) timed on this host for `sample` instances of the same workload."""
    import scipy.linalg as sla
    from oracle import gp_posterior as ogp, cbc as ocbc, socp as osocp, unicycle as ouni
    h = {k: v[:sample].double().cpu().numpy() for k, v in {**p, **task}.items() if v.dim() > 0 and v.shape[0] >= sample}
    sign, relax_mask = task["sign"].double().cpu().numpy(), task["relax_mask"].double().cpu().numpy()
    Kp, tw, gammas = (task[k].double().cpu().numpy() for k in ("Kp", "tw", "gammas"))
    states = []
    for i in range(sample):     # refit state: not timed (cached in the reference between refits)
        states.append(ogp.refit_state(h["X"][i], h["U"][i], h["Xdot"][i], h["Bm"][i], h["ell"][i], h["s2"][i],
                                      h["M0"][i], h["jitter"][i][None] / 1e-5))
    clf = ouni.CLFCartesian(Kp)
    from threadpoolctl import threadpool_limits
    limiter = threadpool_limits(limits=1)          # a scalar, single-thread port: cores = 1
    t0 = time.perf_counter()
    nopt = 0
    for i in range(sample):
        st = states[i]
        x = h["x"][i]
        Mk, Bk = ogp.posterior_step(st["L"][None], st["alpha"][None], h["X"][i][None], st["UHB"][None],
                                    h["ell"][i][None], h["s2"][i][None], h["Bm"][i][None], h["M0"][i][None], x[None])
        fhat, ghat = ouni.ackermann_f(x), ouni.ackermann_g(x, 4.0)
        plan, dplan = h["plan"][i], h["dot_plan"][i]
        const = clf.grad_clf_wrt_goal(x, plan) @ dplan + 10.0 * clf.clf(x, plan)
        terms = [ocbc.reldeg1_terms(Mk[0], Bk[0], h["A"][i], clf.grad_clf(x, plan), const, fhat, ghat, sign=-1.0)]
        for k in range(2):
            ob = ouni.ObstacleCBF(h["centers"][i, k], h["radii"][i, k], tuple(tw))
            terms.append(ocbc.reldeg1_terms(Mk[0], Bk[0], h["A"][i], ob.grad_cbf(x), gammas[k] * ob.cbf(x), fhat, ghat))
        cones = [ocbc.convert_cbc_terms_to_socp_terms(*tm, 0) for tm in terms]
        sol = osocp.clf_cbf_socp(h["w"][i], h["r"][i], cones, h["rho"][i], relax_mask)
        nopt += sol["status"] == "optimal"
    el = time.perf_counter() - t0
    limiter.restore_original_limits()
    return dict(value=sample / el, unit="control steps/s (instance-steps)", cores=1,
                kind="port",
                sample="%d instances of the same N=%d,n=%d,m=%d workload, one at a time, factor cached "
                       "(numpy/scipy oracle, BLAS limited to 1 thread: triangular solve + closed-form terms + coneqp), %.1f s; "
                       "host has %d hardware threads" % (sample, N, n, m, el, os.cpu_count()))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (one-GPU boxes): BCBF_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0, BCBF_BENCH_BACKEND=gloo
    # replaces RCCL -- the N>1 code path is then exercised without a second GPU
    if os.environ.get("BCBF_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("BCBF_BENCH_BACKEND", "nccl")
    # BCBF_BENCH_FORCE_DIST=1: run the N>1 code path (process group, barriers, the final reductions over RCCL) with a
    # single rank too -- the only way to exercise RCCL itself on a one-GPU box
    multi = world > 1 or os.environ.get("BCBF_BENCH_FORCE_DIST") == "1"
    if multi:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    from bayesian_cbf_amd import ops
    from bayesian_cbf_amd.synthetic import make_instances, make_unicycle_task

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    Bt, N, n, m = args.batch, args.ntrain, 3, 2
    K = 3
    shared = args.regime == "shared"
    Bgp = 1 if shared else Bt                # GP instances held by this rank
    p = make_instances(Bgp, N, n, m, dtype=dtype, device=dev, seed=1234 + rank, variant=args.variant)
    task = make_unicycle_task(Bt, dtype=dtype, device=dev, seed=99 + rank)
    # ---- refit (not timed: once per refit, cached between control steps in the reference)
    jit = p["jitter"]
    for attempt in range(4):               # make_psd's retry (control_affine_model.py:899-921): x10 jitter where a pivot failed
        Lop, UHB, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], jit)
        bad = info != 0
        if not bool(bad.any()):
            break
        jit = torch.where(bad[:, None], jit * 10, jit).contiguous()
    assert int((info != 0).sum()) == 0, "Cholesky failed on the synthetic workload after 4 jitter levels"
    Vw, _ = ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False)
    torch.cuda.synchronize()

    # ---- the batch is processed as `chunks` independent sub-batches, each on its own HIP stream:
    # instances never interact, so sub-batch A's SOCP (latency-bound, few waves) overlaps
    # sub-batch B's posterior kernel (HBM-bound) -- also across consecutive steps.
    S = max(1, args.chunks)
    assert Bt % S == 0
    Bc = Bt // S
    dt_plant, L_true, L_mean = 1e-3, 1.0, 4.0
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]

    class Chunk:
        def __init__(self, c):
            sl = slice(c * Bc, (c + 1) * Bc)
            gsl = slice(0, 1) if shared else sl
            q = {k: v[gsl] for k, v in p.items()}
            self.gp = dict(Lop=Lop[gsl], Vw=Vw[gsl], X=q["X"], UHB=UHB[gsl], ell=q["ell"], s2=q["s2"], Bm=q["Bm"],
                           M0=q["M0"], A=q["A"])
            self.task = {k: (v[sl] if v.dim() > 0 and v.shape[0] == Bt else v) for k, v in task.items()}
            self.x = task["x"][sl].clone()
            self.ws = ops.control_workspace(Bc, 2, dtype, dev)
            # one host call per step: constraints -> posterior -> terms -> SOCP -> plant step, on the current stream
            self._step = ops.unicycle_control_step_prepare(self.gp, self.task, self.ws, self.x, dt=dt_plant, L_true=L_true,
                                                           L_mean=L_mean, clf_gamma=10.0, max_iters=20)

        def step(self, ev0=None, ev1=None):
            self._step(ev0, ev1)

    chunks = [Chunk(c) for c in range(S)]
    torch.cuda.synchronize()

    def step(ev=None):
        for c, ch in enumerate(chunks):
            with torch.cuda.stream(streams[c]):
                if ev is None:
                    ch.step()
                else:
                    ch.step(ev[c][0], ev[c][1])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    def barrier():
        if multi:
            import torch.distributed as dist
            dist.barrier()

    # ---- timed region: exactly `steps` steps, HIP events around the dominant kernel (on its stream)
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(S)]
          for _ in range(args.steps)]
    for row in ev:                      # instantiate the hipEvent handles (torch creates them on first record)
        for c, (e0, e1) in enumerate(row):
            e0.record(streams[c])
            e1.record(streams[c])
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(ev[s])
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for row in ev for a, b in row]))
    status = torch.cat([ch.ws["status"] for ch in chunks])
    iters = torch.cat([ch.ws["iters"] for ch in chunks])

    n_opt = int((status == 0).sum())
    stats = torch.tensor([elapsed, float(n_opt), float(Bt), float(iters.float().mean())], dtype=torch.float64, device=dev)
    if multi:
        import torch.distributed as dist
        if backend != "nccl":
            stats = stats.cpu()
        tmax = stats[:1].clone()
        sums = stats[1:3].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)            # the final (only) reduction: a few numbers over RCCL
        stats[1:3] = sums
        elapsed = float(tmax[0])
    total_instances = float(stats[2])
    ms_per_step = elapsed / args.steps * 1e3
    value = total_instances * args.steps / elapsed

    if rank == 0 and shared:
        # regime S: the factor is cache resident; the kernel is a triangular solve with (1+m) right-hand sides per
        # query on the matrix cores: N^2 flop per column (N^2/2 multiply-adds) -- Gram / mean accumulation not counted
        flops_launch = float(Bc) * (1 + m) * N * N
        achieved = flops_launch / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "control steps/sec (GP posterior + CBF-QP) at N_train=%d, batch=%d; shared learned model" % (N, Bt),
            "value": value, "unit": "control steps/s (instance-steps: batch x batched steps/s)",
            "batched_steps_per_s": args.steps / elapsed, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "unicycle x in R^3, u in R^2: GP posterior + 3 chance constraints + SOCP per step, "
                                   "ONE learned model queried by every closed loop (Monte-Carlo rollouts)",
                       "N_train": N, "state_dim": n, "ctrl_dim": m, "batch_per_gpu": Bt, "constraints": K,
                       "regime": "shared GP (S)", "inputs": args.variant, "streams": S,
                       "parallelism": "closed loops sharded, dp%d" % world},
            "solver": {"optimal_fraction": float(stats[1]) / total_instances, "mean_iters": float(stats[3])},
            "roofline": {"bound": "mfma", "kernel": "posterior_shared_kernel" if args.dtype == "f32" and N <= 1536 else
                         "posterior_step_kernel (cache-resident factor, VALU; the matrix-core kernel is fp32, N <= ~1600)",
                         "achieved": achieved,
                         "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F32_MFMA_PEAK_TFLOPS,
                         "traffic": None, "kernel_ms": kern_ms, "algorithmic_flops_per_launch": flops_launch,
                         "queries_per_launch": Bc},
        }
        print(json.dumps(out))
    elif rank == 0:
        bytes_launch = algorithmic_bytes_per_instance(N, n, m, p["X"].element_size()) * Bc
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "control steps/sec (GP posterior + CBF-QP) at N_train=%d, batch=%d; HBM GB/s vs peak" % (N, Bt),
            "value": value,
            "unit": "control steps/s (instance-steps: batch x batched steps/s)",
            "batched_steps_per_s": args.steps / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "unicycle x in R^3, u in R^2: GP posterior + 3 chance constraints (1 CLC + 2 obstacle "
                                   "CBCs) + SOCP per step, independent GP per instance",
                       "N_train": N, "state_dim": n, "ctrl_dim": m, "batch_per_gpu": Bt, "constraints": K,
                       "regime": "independent GPs (I)", "inputs": args.variant, "streams": S, "parallelism": "instances sharded, dp%d" % world},
            "solver": {"optimal_fraction": float(stats[1]) / total_instances, "mean_iters": float(stats[3])},
            "roofline": {"bound": "hbm", "kernel": "posterior_step_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(N, Bc, args.dtype, bytes_launch),
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch, "instances_per_launch": Bc},
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(p, task, args.cpu_sample, N, n, m)
        print(json.dumps(out))
    if multi:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
