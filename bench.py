#!/usr/bin/env python3
"""Headline benchmark: batched GP-posterior + CBF/CLF chance-constraint + SOCP control steps.

Workload = BASELINE.json configs[2]: unicycle (x in R^3, u in R^2), N_train = 512,
batch = 4096 independent control-loop instances per GPU (regime I: every instance owns its GP),
fp32.  One "step" = one pass of the hot path over the batch (every instance takes one control step):
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

# The default schedule runs four part batches on four HIP streams; ROCm maps streams onto GPU_MAX_HW_QUEUES (default 4)
# hardware queues per process, the null stream included, and two streams that share a queue serialize (4 parts at the
# default: 0.446 ms per step; with 8 queues: 0.309).  A runtime setting read when HIP initialises: set it before torch
# loads, unless the caller already chose.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
DEFAULT_STEPS, DEFAULT_WARMUP = 200, 60   # no flags: long enough that clocks / caches / the two streams' pipeline are in
                                          # their steady state (with 5 untimed steps a 20-step run measures ~5 % slower);
                                          # EXACTLY --warmup untimed steps run before the timed region, never more
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense (MI355X_MICROARCH.md)
F64_MFMA_PEAK_TFLOPS = 78.6   # v_mfma_f64_16x16x4_f64, dense (MI355X_MICROARCH.md)
ACHIEVED_METHOD = ("algorithmic units of ALL launches of the kernel in the timed region / time during which at least one of "
                   "them was executing (HIP events around every launch, on its stream); kernel_ms = mean duration of one "
                   "launch, as a kernel trace reports it -- with launches_per_step > 1 the launches overlap one another; "
                   "algorithmic_*_per_launch / instances_per_launch are means over the part batches (sizes: *_by_part)")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=DEFAULT_STEPS)
    ap.add_argument("--warmup", type=int, default=DEFAULT_WARMUP)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--ntrain", type=int, default=512)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--variant", default="dense")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="instances timed for the CPU baseline (0 = skip)")
    ap.add_argument("--parts", type=int, default=0,
                    help="part batches, each on its own HIP stream (ops.ConcurrentControlLoop): one part's solve launch "
                         "runs beside the other parts' posterior streams; 1 = the whole batch on one stream.  0 (default) = the "
                         "measured optimum: regime I 4 parts when the runtime has >= 6 hardware queues (GPU_MAX_HW_QUEUES, set "
                         "to 8 above) else 3 (ms per step at the BASELINE config: 1 part 0.409, 2: 0.357, 3: 0.318, 4: 0.309 "
                         "/ 0.446 with 4 queues, 5: 0.363); regime S 2 parts")
    ap.add_argument("--regime", choices=["independent", "shared"], default="independent",
                    help="independent: every instance owns its GP (headline, HBM bound); shared: one learned model, "
                         "`batch` closed loops (Monte-Carlo rollouts, BASELINE configs[3]; matrix-core bound)")
    ap.add_argument("--config", choices=["c3", "c4", "c5", "learn"], default="c3",
                    help="c3 (default): this file's headline workload.  c4 / c5: the Monte-Carlo rollouts / online growth "
                         "harnesses (examples_mc_rollouts.py, tools/bench_online.py) with the same --gpus N launcher: "
                         "EVERY other flag on the command line (also --steps / --batch / --dtype ...) goes to that harness "
                         "untouched and means what the harness says it means")
    # c4 / c5: only --config and --gpus are this file's; the rest of the command line is the harness's own (a shared
    # flag name such as --steps or --batch must not be swallowed here and replaced by the harness default)
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--config", choices=["c3", "c4", "c5", "learn"], default="c3")
    pre.add_argument("--gpus", type=int, default=1)
    pargs, rest = pre.parse_known_args()
    if pargs.config != "c3":
        pargs.rest = rest
        return pargs
    args = ap.parse_args()
    args.rest = []
    return args


def run_other_config(args):
    """--config c4 | c5: hand over to that config's harness IN this process (it starts its own ranks for --gpus N, or
    joins the launcher's job), with its BASELINE.json size as the default."""
    import runpy
    if args.config == "c4":
        script = os.path.join(ROOT, "examples_mc_rollouts.py")
        argv = ["--gpus", str(args.gpus)] + (args.rest or ["--trajectories", "32768", "--steps", "200", "--graph"])
    elif args.config == "learn":
        script = os.path.join(ROOT, "tools", "bench_learning_loop.py")
        argv = ["--gpus", str(args.gpus)] + args.rest
    else:
        script = os.path.join(ROOT, "tools", "bench_online.py")
        argv = ["--gpus", str(args.gpus)] + args.rest
    sys.argv = [script] + argv
    runpy.run_path(script, run_name="__main__")


def kernel_source_hash():
    """sha256 of the roofline kernel's sources (what tools/collect_profiles.py stores beside the counters)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("bayesian_cbf_amd/csrc/posterior_step.hip", "bayesian_cbf_amd/csrc/bcbf_common.h"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()


def measured_traffic(N, Bt, dtype_name, bytes_launch):
    """HBM bytes per launch of the roofline kernel from the committed PMC passes (profiles/*_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 corrections applied).  A counter run cannot
    be nested inside this process, so the number is attached only when the profiled workload is this workload AND the
    kernel source still hashes to what the counters were taken from (`kernel_source.sha256` in the JSON): a kernel change
    that could break the 1.004 x then shows as `traffic: null` until the PMC passes are rerun (tools/run_profiles.sh)."""
    best, src = None, None
    cur = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        w = d.get("workload", {})
        if w.get("N_train") == N and w.get("dtype") == dtype_name and w.get("batch"):
            want = d.get("kernel_source", {}).get("sha256")
            if want is not None:
                cur = cur or kernel_source_hash()
                if want != cur:
                    best, src = None, "%s is STALE: the kernel source changed since its counter passes (rerun tools/run_profiles.sh)" % os.path.relpath(path, ROOT)
                    continue
            best = d["hbm_bytes_per_launch"] * (Bt / float(w["batch"]))    # per-instance traffic is batch independent
            src = "%s (rocprofv3 --pmc passes of this workload, committed%s; NOT measured in this run)" % (
                os.path.relpath(path, ROOT), ", kernel source hash checked" if want else "")
    return best, src


def algorithmic_bytes_per_instance(N, n, m, itemsize):
    """packed factor + whitened targets + train inputs + UH*B, each read once."""
    return itemsize * (N * (N + 1) // 2 + N * n + N * n + N * (1 + m))


def effective_cpus():
    """(cores this process may actually use, how): the smallest of the hardware thread count, the scheduler affinity mask and
    the cgroup CPU quota (a container on a 256-thread host is often given 16 CPUs' worth of run time: 64 busy processes then
    each run at a quarter speed -- measured on this pool: 64 workers x 115/s against 457/s for one alone)."""
    n, how = os.cpu_count() or 1, "os.cpu_count"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, how = a, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse is not None:
                q, per = parse(txt)
            else:
                q, per = txt, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if q not in ("max", "-1"):
                c = max(1, int(float(q) / float(per) + 0.5))
                if c < n:
                    n, how = c, "cgroup cpu quota (%s)" % path
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, how


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_CPU_SHARED = {}


def _cpu_worker(lo, hi, reps, barrier, queue):
    """One worker PROCESS of the process-per-core CPU baseline: the reference-style scalar loop (one instance per call,
    one thread) over its slice of the sample.  Refit (cached upstream) before the barrier, untimed."""
    from threadpoolctl import threadpool_limits
    from oracle import control_step as ostep, gp_posterior as ogp
    import torch as _t
    _t.set_num_threads(1)
    h = _CPU_SHARED["h"]
    with threadpool_limits(limits=1):
        states = [ogp.refit_state(h["X"][i], h["U"][i], h["Xdot"][i], h["Bm"][i], h["ell"][i], h["s2"][i], h["M0"][i],
                                  h["jitter"][i][None] / 1e-5) for i in range(lo, hi)]
        barrier.wait()
        t0 = time.time()
        for _ in range(reps):
            for i in range(lo, hi):
                st = states[i - lo]
                Mk, Bk = ogp.posterior_step(st["L"][None], st["alpha"][None], h["X"][i][None], st["UHB"][None],
                                            h["ell"][i][None], h["s2"][i][None], h["Bm"][i][None], h["M0"][i][None],
                                            h["x"][i][None])
                ostep.control_step(h["x"][i], h["plan"][i], h["dot_plan"][i], Mk[0], Bk[0], h["A"][i], h["Kp"], 10.0,
                                   h["centers"][i], h["radii"][i], h["tw"], h["gammas"], 4.0, h["w"][i], h["r"][i],
                                   h["rho"][i], h["relax_mask"])
        queue.put((t0, time.time(), (hi - lo) * reps))


def cpu_baseline(sample, N, n, m, seed=1234, task_seed=99):
    """The oracle (a numpy / torch-CPU port of the reference's arithmetic, Cholesky factor cached as the reference caches
    it between refits) timed on this host on a bounded sample of the same workload -- the same generators and seeds as
    the GPU run, drawn on the CPU -- four ways: (a) reference style, one instance at a time, one thread;
    (b) vectorised over the batch, one thread; (c) vectorised, 16 threads (more threads are slower, see below);
    (d) the reference-style loop as one PROCESS per core (up to 64 workers, one thread each, a slice of the sample each).
    `value` is the fastest of them, `cores` the threads / processes it used.
    MUST run before this process's first GPU call: (d) forks workers, and nothing that has initialised the GPU may
    spawn or replace a process on this pool -- main() calls it first and it refuses otherwise."""
    from threadpoolctl import threadpool_limits
    from bayesian_cbf_amd.distributed import open_gpu_descriptors
    from bayesian_cbf_amd.synthetic import make_instances, make_unicycle_task
    from oracle import batched as ob, control_step as ostep, gp_posterior as ogp
    L_mean = 4.0
    ncpu = os.cpu_count() or 1
    gpu_fds = open_gpu_descriptors()
    variants = []
    f64 = torch.float64
    p = make_instances(sample, N, n, m, dtype=f64, device="cpu", seed=seed)
    task = make_unicycle_task(sample, dtype=f64, device="cpu", seed=task_seed)
    take = lambda v, k: v[:k] if (v.dim() > 0 and v.shape[0] >= k and v.shape[0] == sample) else v

    # ---- (d) process per core: reference-style scalar loop, workers forked BEFORE anything touches the GPU
    usable, usable_how = effective_cpus()
    workers = min(usable, 64, sample)
    if gpu_fds:
        variants.append(dict(name="scalar loop, one process per core: SKIPPED (this process already holds %s)" % gpu_fds[0],
                             cores=0, value=0.0, instances=0, seconds=0.0))
    elif workers > 1:
        import multiprocessing as mp
        ctx = mp.get_context("fork")
        # at the usable-CPU count, and at 16 as well when more are reported: a CPU quota this process cannot see shows up
        # as 64 workers running at a quarter speed each, and then 16 is the honest best
        for nw in sorted({min(workers, 16), workers}):
            per = max(1, min(32, sample // nw))
            sd = nw * per
            _CPU_SHARED["h"] = {k: take(v, sd).numpy() for k, v in {**p, **task}.items()}
            barrier, queue = ctx.Barrier(nw), ctx.Queue()
            reps = max(1, int(round(1200 / per)))            # ~1200 instance-steps per worker: a few seconds each
            procs = [ctx.Process(target=_cpu_worker, args=(w * per, (w + 1) * per, reps, barrier, queue)) for w in range(nw)]
            for pr in procs:
                pr.start()
            res = []
            try:                                   # (a worker that never reports -- a fork of a threaded parent can in principle
                for _ in procs:                    #  inherit a held lock -- must not hang the benchmark: bounded wait, then give up)
                    res.append(queue.get(timeout=600))
            except Exception:
                res = None
            for pr in procs:
                if res is None and pr.is_alive():
                    pr.terminate()
                pr.join(30)
            if res is None:
                variants.append(dict(name="scalar loop, one process per core: ABANDONED (a worker did not report within 600 s)",
                                     cores=0, value=0.0, instances=0, seconds=0.0))
                _CPU_SHARED.clear()
                continue
            el = max(r[1] for r in res) - min(r[0] for r in res)
            done = sum(r[2] for r in res)
            pw = sorted(r[2] / (r[1] - r[0]) for r in res)
            variants.append(dict(name="scalar loop (reference style), one process per core, one thread each", cores=nw,
                                 value=done / el, instances=sd, passes=reps, seconds=el,
                                 per_worker_value=dict(min=pw[0], median=pw[len(pw) // 2], max=pw[-1])))
            _CPU_SHARED.clear()

    # ---- (a) scalar loop, 1 thread
    sa = min(sample, 2048)
    h = {k: take(v, sa).numpy() for k, v in {**p, **task}.items()}
    states = [ogp.refit_state(h["X"][i], h["U"][i], h["Xdot"][i], h["Bm"][i], h["ell"][i], h["s2"][i], h["M0"][i],
                              h["jitter"][i][None] / 1e-5) for i in range(sa)]      # refit: not timed (cached upstream)
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        for i in range(sa):
            st = states[i]
            Mk, Bk = ogp.posterior_step(st["L"][None], st["alpha"][None], h["X"][i][None], st["UHB"][None],
                                        h["ell"][i][None], h["s2"][i][None], h["Bm"][i][None], h["M0"][i][None],
                                        h["x"][i][None])
            ostep.control_step(h["x"][i], h["plan"][i], h["dot_plan"][i], Mk[0], Bk[0], h["A"][i], h["Kp"], 10.0,
                               h["centers"][i], h["radii"][i], h["tw"], h["gammas"], L_mean, h["w"][i], h["r"][i],
                               h["rho"][i], h["relax_mask"])
        el = time.perf_counter() - t0
    variants.append(dict(name="scalar loop (reference style: one instance per call)", cores=1, value=sa / el,
                         instances=sa, seconds=el))
    del states

    # ---- (b), (c) vectorised over the batch (torch CPU: batched triangular solve + the batched cone solver)
    sv = min(sample, 1024)
    q = {k: take(v, sv) for k, v in {**p, **task}.items()}
    Ls, Vws, UHBs = [], [], []
    for c0 in range(0, sv, 128):                                # refit in slices: not timed
        sl = slice(c0, min(c0 + 128, sv))
        L_, Vw_, UHB_ = ob.refit(q["X"][sl], q["UH"][sl], q["Xdot"][sl], q["Bm"][sl], q["ell"][sl], q["s2"][sl],
                                 q["M0"][sl], q["jitter"][sl])
        Ls.append(L_); Vws.append(Vw_); UHBs.append(UHB_)
    L_, Vw_, UHB_ = torch.cat(Ls), torch.cat(Vws), torch.cat(UHBs)
    del Ls, Vws, UHBs
    # torch's CPU kernels on [B, 512, 512] batches stop scaling (and collapse under oversubscription: 256 threads took
    # 240 s for the pass one thread does in 0.4 s on the EPYC 9575F host): the multi-thread variant uses 16
    many = min(ncpu, usable, 16)
    for threads in ((1, many) if many > 1 else (1,)):
        torch.set_num_threads(threads)
        reps = 1 if threads == 1 else 2
        with threadpool_limits(limits=threads):
            t0 = time.perf_counter()
            for _ in range(reps):
                ob.control_step(L_, Vw_, q["X"], UHB_, q["ell"], q["s2"], q["Bm"], q["M0"], q["A"], q["x"], q["plan"],
                                q["dot_plan"], q["Kp"], 10.0, q["centers"], q["radii"], q["tw"], q["gammas"], L_mean,
                                q["w"], q["r"], q["rho"], q["relax_mask"])
            el = (time.perf_counter() - t0) / reps
        variants.append(dict(name="vectorised over the batch (torch CPU)", cores=threads, value=sv / el, instances=sv,
                             seconds=el))
    torch.set_num_threads(ncpu)
    best = max(variants, key=lambda v: v["value"])
    return dict(value=best["value"], unit="control steps/s (instance-steps)", cores=best["cores"], kind="port",
                sample="%s; %d instances of the same N=%d,n=%d,m=%d workload (same generator and seeds, drawn on the CPU) per "
                       "pass, factor cached; fp64; measured before this process's first GPU call; host: %s, %d hardware threads"
                       % (best["name"], best["instances"], N, n, m, cpu_model(), ncpu),
                cpu_model=cpu_model(), host_threads=ncpu, usable_cpus=usable, usable_cpus_how=usable_how, variants=variants)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: N ranks (one process per GPU) under torch.distributed.run as a CHILD
    process; this parent never touches the GPU (bayesian_cbf_amd.distributed.launch_ranks)."""
    from bayesian_cbf_amd.distributed import launch_ranks as _launch
    return _launch(os.path.abspath(__file__), sys.argv[1:], args.gpus)


def main():
    args = parse()
    if args.config != "c3":
        return run_other_config(args)
    # BCBF_BENCH_FORCE_LAUNCH=1 (test hook): also a one-GPU run goes through the launcher parent -> child rank path
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("BCBF_BENCH_FORCE_LAUNCH") == "1"):
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and "WORLD_SIZE" in os.environ:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is used\n" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the CPU baseline (rank 0 of a one-GPU run) is measured FIRST: its process-per-core variant forks workers, which must
    # happen before this process initialises the GPU
    cpu_base = None
    if world == 1 and rank == 0 and args.cpu_sample > 0 and args.regime == "independent":
        cpu_base = cpu_baseline(args.cpu_sample, args.ntrain, 3, 2, seed=1234 + rank, task_seed=99 + rank)
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
    # ---- GP state buffers.  The refit that fills them (not timed: once per refit, cached between control steps in the
    # reference) runs further down, AFTER the host-side preparation of the loop and DIRECTLY before the warm-up steps:
    # the device lowers its clocks within a few ms of idleness and needs ~15 ms of load to raise them again
    # (tools/probe_ramp.py: the first ~40 steps after an idle gap run 6 % slow; behind the refit the first step is at
    # the steady rate), so the order "prepare the host side, then refit, then W warm-up steps, then K timed steps"
    # measures the steady rate without a single extra launch.
    f = dict(dtype=dtype, device=dev)
    Lop = torch.empty(Bgp, ops.lop_elems(N, dtype), **f)
    UHB = torch.empty(Bgp, N, 1 + m, **f)
    Vw = torch.empty(Bgp, N, n, **f)
    info = torch.empty(Bgp, dtype=torch.int32, device=dev)

    def device_setup():
        jit = p["jitter"]
        for attempt in range(4):           # make_psd's retry (control_affine_model.py:899-921): x10 jitter where a pivot failed
            ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], jit, out=(Lop, UHB, info))
            bad = info != 0
            if not bool(bad.any()):
                break
            jit = torch.where(bad[:, None], jit * 10, jit).contiguous()
        assert int((info != 0).sum()) == 0, "Cholesky failed on the synthetic workload after 4 jitter levels"
        ops.potrs(Lop, p["Xdot"], p["UH"], p["M0"], want_alpha=False, out_Vw=Vw)

    # ---- one step = every instance takes one control step: posterior -> task rows + terms + SOCP -> plant step.
    # Default schedule: the batch is split into `parts` part batches (instances never interact), each on its own HIP
    # stream; the device overlaps one part's latency-bound solve launch with another part's HBM-bound posterior stream.
    try:
        hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        hwq = 4
    S = args.parts if args.parts > 0 else (2 if shared else (4 if hwq >= 6 else 3))
    base_, rem_ = divmod(Bt, S)
    Bcs = [base_ + (1 if c < rem_ else 0) for c in range(S)]         # instances per part batch (4096 -> 1366 + 1365 + 1365)
    Bc = Bt / S                                                      # mean instances per launch
    dt_plant, L_true, L_mean = 1e-3, 1.0, 4.0
    gsl = slice(0, 1) if shared else slice(None)
    gp = dict(Lop=Lop[gsl], Vw=Vw[gsl], X=p["X"][gsl], UHB=UHB[gsl], ell=p["ell"][gsl], s2=p["s2"][gsl],
              Bm=p["Bm"][gsl], M0=p["M0"][gsl], A=p["A"][gsl])
    x = task["x"].clone()
    if S > 1:
        loop = ops.ConcurrentControlLoop(gp, task, x, parts=S, dt=dt_plant, L_true=L_true, L_mean=L_mean, clf_gamma=10.0,
                                         max_iters=20)
        ev_stream = loop.streams
        step, status_t, iters_t = loop.step, loop.status, loop.iters
    else:
        ws = ops.control_workspace(Bt, 2, dtype, dev)
        one = ops.unicycle_control_step_prepare(gp, task, ws, x, dt=dt_plant, L_true=L_true, L_mean=L_mean, clf_gamma=10.0,
                                                max_iters=20)
        ev_stream = [torch.cuda.current_stream(dev)]
        step = (lambda ev=None: one() if ev is None else one(ev[0][0], ev[0][1]))
        status_t, iters_t = ws["status"], ws["iters"]
    torch.cuda.synchronize()

    def barrier():
        if multi:
            import torch.distributed as dist
            dist.barrier()

    # HIP events around every launch of the dominant kernel (on its stream), created and instantiated BEFORE the warm-up
    # so that nothing but the barrier sits between the last untimed step and the first timed one
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(S)]
          for _ in range(args.steps)]
    for row in ev:                      # instantiate the hipEvent handles (torch creates them on first record)
        for c, (e0, e1) in enumerate(row):
            e0.record(ev_stream[c])
            e1.record(ev_stream[c])
    ev_base = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    # ---- device-side setup (refit + whitened targets), then, untimed, exactly the W warm-up steps asked for
    device_setup()
    for st_ in ev_stream:                   # the part batches' streams start behind the refit on the current stream
        st_.wait_stream(torch.cuda.current_stream(dev))
    for _ in range(args.warmup):
        step()
    # ---- timed region: exactly `steps` steps, bracketed by barrier + synchronize on both sides
    barrier()
    torch.cuda.synchronize()
    ev_base.record(ev_stream[0])
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(ev[s])
    submitted = time.perf_counter() - t0      # host side: all launches of the timed region handed to the HIP runtime
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # per-launch duration (what a kernel trace reports), and the time during which AT LEAST ONE launch of the kernel
    # was executing (with one stream the two coincide; with concurrent parts the launches overlap one another, a
    # launch's own duration then includes the time it shared the device, and bytes / union time is the rate the
    # kernel sustained while it ran)
    spans = sorted((ev_base.elapsed_time(a), ev_base.elapsed_time(b)) for row in ev for a, b in row)
    kern_ms = float(np.mean([b_ - a_ for a_, b_ in spans]))
    busy_ms, cur_a, cur_b = 0.0, spans[0][0], spans[0][1]
    for a_, b_ in spans[1:]:
        if a_ > cur_b:
            busy_ms += cur_b - cur_a
            cur_a, cur_b = a_, b_
        else:
            cur_b = max(cur_b, b_)
    busy_ms += cur_b - cur_a
    n_launch = len(spans)
    status, iters = status_t, iters_t
    # where inside the timed region the time went (device time stamps of the step ends): a short run that starts from few
    # untimed steps still contains the ramp of the clocks -- reported, never removed from `value`
    ends = [max(ev_base.elapsed_time(b_) for _, b_ in row) for row in ev]
    per = np.diff(np.array([0.0] + ends))
    k5 = max(1, min(5, len(per) // 2))
    if os.environ.get("BCBF_BENCH_DUMP"):            # development: per-step device times of the timed region
        json.dump(dict(per=[round(float(v), 4) for v in per], spans=[[round(a_, 4), round(b_, 4)] for a_, b_ in spans[:64]]),
                  open(os.environ["BCBF_BENCH_DUMP"], "w"))
    region = {"first_steps_ms": float(np.mean(per[:k5])), "last_steps_ms": float(np.mean(per[-k5:])), "averaged_over": k5,
              "host_submit_ms_per_step": submitted / args.steps * 1e3}

    n_opt = int((status == 0).sum())
    stats = torch.tensor([elapsed, float(n_opt), float(Bt), float(iters.float().mean())], dtype=torch.float64, device=dev)
    rank_ms, comm_world = [elapsed / args.steps * 1e3], 1
    if multi:
        import torch.distributed as dist
        if backend != "nccl":
            stats = stats.cpu()
        tmax = stats[:1].clone()
        sums = stats[1:3].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)            # the final (only) reduction: a few numbers over RCCL
        comm_world = dist.get_world_size()
        per_rank = [torch.zeros_like(stats[:1]) for _ in range(comm_world)]
        dist.all_gather(per_rank, stats[:1].clone())           # every rank's own wall time (reporting only)
        rank_ms = [float(t[0]) / args.steps * 1e3 for t in per_rank]
        stats[1:3] = sums
        elapsed = float(tmax[0])
    schedule = ("%d part batches on %d streams (posterior launch, then solve launch, per part): one part's solve runs "
                "beside another part's posterior" % (S, S)) if S > 1 else "one stream: posterior launch, then solve launch"
    comm = {"backend": ("rccl" if backend == "nccl" else backend) if multi else None, "world_size": comm_world,
            "per_rank_ms_per_step": rank_ms}
    total_instances = float(stats[2])
    ms_per_step = elapsed / args.steps * 1e3
    value = total_instances * args.steps / elapsed

    if rank == 0 and shared:
        # regime S: the factor is cache resident; the kernel is a triangular solve with (1+m) right-hand sides per
        # query on the matrix cores: N^2 flop per column (N^2/2 multiply-adds) -- Gram / mean accumulation not counted
        flops_launch = float(Bc) * (1 + m) * N * N
        mfma_peak = F64_MFMA_PEAK_TFLOPS if args.dtype == "f64" else F32_MFMA_PEAK_TFLOPS
        if N <= 512 and n <= 4:
            shared_kernel = "posterior_shared_reg_kernel<%s>" % ("double" if args.dtype == "f64" else "float")
        elif args.dtype == "f32" and N <= 1536:
            shared_kernel = "posterior_shared_kernel"
        else:
            shared_kernel = ("posterior_step_kernel (cache-resident factor, VALU; the matrix-core kernels hold N <= 512 with the "
                             "solution in registers, N <= ~1600 in fp32 with it in LDS)")
        achieved = flops_launch * n_launch / (busy_ms * 1e-3) / 1e12
        out = {
            "metric": "control steps/sec (GP posterior + CBF-QP) at N_train=%d, batch=%d; shared learned model" % (N, Bt),
            "value": value, "unit": "control steps/s (instance-steps: batch x batched steps/s)",
            "batched_steps_per_s": args.steps / elapsed, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "unicycle x in R^3, u in R^2: GP posterior + 3 chance constraints + SOCP per step, "
                                   "ONE learned model queried by every closed loop (Monte-Carlo rollouts)",
                       "N_train": N, "state_dim": n, "ctrl_dim": m, "batch_per_gpu": Bt, "constraints": K,
                       "regime": "shared GP (S)", "inputs": args.variant, "schedule": schedule,
                       "parallelism": "closed loops sharded, dp%d" % world},
            "solver": {"optimal_fraction": float(stats[1]) / total_instances, "mean_iters": float(stats[3]),
                       # `value` counts every instance-step the hot path ran (posterior + terms + solve attempt); this is
                       # the share of them whose program was solved (the rest are infeasible: the oracle agrees)
                       "solved_instance_steps_per_s": value * float(stats[1]) / total_instances},
            "comm": comm,
            "timed_region": region,
            "roofline": {"bound": "mfma", "kernel": shared_kernel,
                         "achieved": achieved,
                         "peak": mfma_peak, "unit": "TFLOP/s", "frac": achieved / mfma_peak,
                         "traffic": None, "kernel_ms": kern_ms, "kernel_busy_ms_per_step": busy_ms / args.steps,
                         "launches_per_step": S, "achieved_method": ACHIEVED_METHOD,
                         "algorithmic_flops_per_launch": flops_launch, "queries_per_launch": Bc,
                         "queries_per_launch_by_part": Bcs},
        }
        print(json.dumps(out))
    elif rank == 0:
        bytes_launch = algorithmic_bytes_per_instance(N, n, m, p["X"].element_size()) * Bc
        traffic, traffic_src = measured_traffic(N, Bc, args.dtype, bytes_launch)
        achieved = bytes_launch * n_launch / (busy_ms * 1e-3) / 1e9
        # the same kernel on the same part batch with the device to itself (after the timed region): one launch per event
        # pair, nothing else running -- bytes per launch / its own duration, the figure a kernel trace of a one-stream run gives
        alone = {}
        if S > 1:
            for tag, cnt in (("part_batch", Bcs[0]), ("full_batch", Bt)):
                sl = slice(0, cnt)
                pa = [gp[k][sl] for k in ("Lop", "Vw", "X", "UHB", "ell", "s2", "Bm", "M0")] + [x[sl].contiguous()]
                for _ in range(3):
                    ops.posterior_step(*pa)
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
                torch.cuda.synchronize()
                for e0, e1 in evs:
                    e0.record()
                    ops.posterior_step(*pa)
                    e1.record()
                torch.cuda.synchronize()
                ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
                gbs = algorithmic_bytes_per_instance(N, n, m, p["X"].element_size()) * cnt / (ms * 1e-3) / 1e9
                alone[tag] = {"instances": cnt, "kernel_ms": ms, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS}
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
                       "regime": "independent GPs (I)", "inputs": args.variant, "schedule": schedule, "parallelism": "instances sharded, dp%d" % world},
            "solver": {"optimal_fraction": float(stats[1]) / total_instances, "mean_iters": float(stats[3]),
                       # `value` counts every instance-step the hot path ran (posterior + terms + solve attempt); this is
                       # the share of them whose program was solved (the rest are infeasible: the oracle agrees)
                       "solved_instance_steps_per_s": value * float(stats[1]) / total_instances},
            "comm": comm,
            "timed_region": region,
            "roofline": {"bound": "hbm", "kernel": "posterior_step_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": kern_ms, "kernel_busy_ms_per_step": busy_ms / args.steps, "launches_per_step": S,
                         "achieved_method": ACHIEVED_METHOD,
                         "algorithmic_bytes_per_launch": bytes_launch, "instances_per_launch": Bc,
                         "instances_per_launch_by_part": Bcs,
                         "algorithmic_bytes_per_step": algorithmic_bytes_per_instance(N, n, m, p["X"].element_size()) * Bt},
        }
        # the vendor figure AND a read rate measured on this box -- the operator buffer the kernel streams, read once per
        # launch by bcbf_hbm_read_probe (same 16-byte non-temporal loads, no arithmetic), ONE stream, after the timed region.  It is a
        # reference point, not a ceiling: the headline's four staggered part-batch launches have measured a few percent ABOVE this
        # one-stream probe (round 5: 7.29 against 7.02 TB/s), so no "fraction of measured peak" is derived from it.
        probe = ops.hbm_read_probe(Lop, launches=10)
        out["roofline"]["read_probe"] = dict(
            gbs=probe["best_gbs"], mean_gbs=probe["mean_gbs"],
            how="bcbf_hbm_read_probe: read-only pass over the %.2f GB operator buffer on one stream, best of %d launches, device to "
                "itself, after the timed region; NOT an upper bound (overlapping launches on several streams read faster); `peak` / "
                "`frac` are on the vendor figure" % (probe["bytes"] / 1e9, probe["launches"]))
        if alone:
            alone["note"] = ("the same kernel with the device to itself, one launch per HIP-event pair, after the timed region: "
                             "algorithmic bytes per launch / that launch's duration (what a kernel trace of a one-stream run "
                             "reports: profiles/*_bench_parts1_kernel_stats.csv)")
            out["roofline"]["unoverlapped"] = alone
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out))
    if multi:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
