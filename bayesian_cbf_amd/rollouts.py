"""Monte-Carlo safety rollouts (BASELINE config 4) and online-GP growth (config 5) drivers.

Batched counterpart of the reference's `unicycle_bayes_cbf_safe_obstacle` recipe
(unicycle_move_to_pose.py:1887-1928: fixed-kernel Ackermann model, CLFCartesian Kp=[.9,1.5,0],
two obstacles at mid path with weights [.7,.3], gamma 5, max_risk 0.01) run as Bt independent
closed loops from perturbed start states: `sample_generator_trajectory` (sampling.py:68-74) for
every trajectory at once, sharded over GPUs by `distributed.shard_range`, statistics reduced once
at the end."""
import math
import os

import torch

from . import ops
from .cbc2 import cbc1_safety_factor
from .distributed import reduce_rollout_stats
from .planner import PiecewiseLinearPlanner


def unicycle_task_tensors(Bt, x0, xg, dtype, device, term_weights=(0.7, 0.3), cbf_gammas=(5.0, 5.0),
                          Kp=(0.9, 1.5, 0.0), cost_weights=(0.33, 0.33, 0.33), max_risk=0.01):
    f = dict(dtype=dtype, device=device)
    x0, xg = x0.to(**f), xg.to(**f)
    R90 = torch.tensor([[0.0, -1.0], [1.0, 0.0]], **f)
    d = x0[:2] - xg[:2]
    mid = (x0[:2] + xg[:2]) / 2
    centers = torch.stack([mid + R90 @ d / 3, mid - R90 @ d / 3]).expand(Bt, 2, 2).contiguous()
    radii = (d.norm() / 4).expand(Bt, 2).contiguous()
    rho = 0.0 if max_risk == 0.5 else cbc1_safety_factor(max_risk)
    return dict(centers=centers, radii=radii, Kp=torch.tensor(Kp, **f), tw=torch.tensor(term_weights, **f),
                gammas=torch.tensor(cbf_gammas, **f), sign=torch.tensor([-1.0, 1.0, 1.0], **f),
                relax_mask=torch.tensor([1.0, 0.0, 0.0], **f), w=torch.tensor(cost_weights, **f).expand(Bt, 3).contiguous(),
                r=torch.zeros(Bt, 2, **f), rho=torch.full((Bt,), rho, **f))


def monte_carlo_safety_rollouts(Bt, numSteps=200, dt=0.05, gp=None, kernel_diag_A=(1e-2, 1e-2, 1e-2),
                                L_mean=1.0, L_true=12.0, start=(-3.0, -1.0, -math.pi / 4), goal=(0.0, 0.0, math.pi / 4),
                                start_noise=0.05, max_risk=0.01, dtype=torch.float64, device="cuda", seed=0,
                                record=False, max_iters=30, use_graph=False):
    """Run Bt closed loops for numSteps steps.  `gp`: dict from BatchedControlAffineGP.as_dict() (learned
    residual, one GP per trajectory) or None (fixed-kernel model M_k = 0, B_k = I, A = diag(kernel_diag_A)).
    Returns dict(stats..., x_final[Bt,3], traj (if record)).  Collectives: one, at the end.
    use_graph: capture one closed-loop step (plan row gather, the fused control step, the safety bookkeeping) in a HIP
    graph and replay it numSteps times -- the loop is launch bound for small batches (about ten launches per step)."""
    dev = torch.device(device)
    f = dict(dtype=dtype, device=dev)
    gen = torch.Generator(device=dev).manual_seed(seed)
    x0 = torch.tensor(start, **f)
    xg = torch.tensor(goal, **f)
    task = unicycle_task_tensors(Bt, x0, xg, dtype, dev, max_risk=max_risk)
    planner = PiecewiseLinearPlanner(x0, xg, numSteps, dt, frac_time_to_reach_goal=0.95)
    x = (x0 + start_noise * torch.randn(Bt, 3, generator=gen, **f)).contiguous()
    ws = ops.control_workspace(Bt, 2, dtype, dev)
    if gp is None:
        A = torch.diag(torch.tensor(kernel_diag_A, **f)).expand(Bt, 3, 3).contiguous()
        ws["Mk"].zero_()                                   # fixed-kernel model: M_k = 0, B_k = I are inputs of the step
        ws["Bk"].copy_(torch.eye(3, **f).expand(Bt, 3, 3))
        fixed = dict(A=A)
    min_h = torch.full((Bt,), float("inf"), **f)
    cost = torch.zeros(Bt, **f)
    fails = torch.zeros(Bt, dtype=torch.int32, device=dev)
    gam = task["gammas"]
    traj = torch.empty(numSteps + 1, Bt, 3, **f) if record else None
    if record:
        traj[0] = x
    # the whole plan goes to the device once; per step two broadcast copies into the task buffers
    plan_all = torch.stack([planner.plan(t).to(dtype=dtype) for t in range(numSteps)]).to(dev)
    dplan_all = torch.stack([planner.dot_plan(t).to(dtype=dtype) for t in range(numSteps)]).to(dev)
    task["plan"], task["dot_plan"] = torch.empty(Bt, 3, **f), torch.empty(Bt, 3, **f)
    step = ops.unicycle_control_step_prepare(gp if gp is not None else fixed, task, ws, x, dt=dt, L_true=L_true,
                                             L_mean=L_mean, max_iters=max_iters)
    import time
    w_cost = task["w"]

    def one_step(t):
        # plan row t -> task buffers (t is a python int, or a device index tensor inside the captured graph)
        if torch.is_tensor(t):
            task["plan"].copy_(plan_all.index_select(0, t))
            task["dot_plan"].copy_(dplan_all.index_select(0, t))
        else:
            task["plan"].copy_(plan_all[t])
            task["dot_plan"].copy_(dplan_all[t])
        step()       # one host call, two launches (one for the fixed-kernel model): rows -> terms -> SOCP -> plant step
        # safety bookkeeping in ONE launch: min_h over the obstacle rows h_k(x_t) = cst_k / gamma_k (before the step; a
        # non-finite h counts as a collision), and -- only where the program was solved: an unsolved program (MAXITER /
        # infeasible / bad cone) is where the reference raises ValueError (unicycle_move_to_pose.py:954-964), the kernel
        # leaves that instance's state untouched for the step and its y is not a control -- the cost; else a failure count
        ops.rollout_stats(ws["cst"], ws["y"], ws["status"], w_cost, gam, min_h, cost, fails)

    torch.cuda.synchronize(dev)
    graph = None
    if use_graph and not record:
        tctr = torch.zeros(1, dtype=torch.long, device=dev)
        side = torch.cuda.Stream(device=dev)
        saved = [v.clone() for v in (x, min_h, cost, fails)]
        with torch.cuda.stream(side):                   # warm-up on the capture stream (allocator, lazy module load)
            one_step(tctr)
        side.synchronize()
        for dst, src in zip((x, min_h, cost, fails), saved):
            dst.copy_(src)
        tctr.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            one_step(tctr)
            tctr.add_(1)
        for dst, src in zip((x, min_h, cost, fails), saved):    # capture does not execute, but keep the state explicit
            dst.copy_(src)
        tctr.zero_()
        torch.cuda.synchronize(dev)
    t_loop = time.perf_counter()
    for t in range(numSteps):
        if graph is not None:
            graph.replay()
        else:
            one_step(t)
        if record:
            traj[t + 1] = x
    torch.cuda.synchronize(dev)
    t_loop = time.perf_counter() - t_loop
    collided = ~(min_h >= 0)                       # NaN-safe: anything that is not provably >= 0 is a collision
    stats = reduce_rollout_stats(collided.sum(), min_h.min(), cost.sum() / numSteps, (fails > 0).sum(), Bt)
    dist_to_goal = (x[:, :2] - xg[:2]).norm(dim=1)
    return dict(stats=stats, x_final=x, min_h=min_h, dist_to_goal=dist_to_goal, traj=traj, loop_seconds=t_loop)


def online_pass_bytes(N, n, m, itemsize):
    """Algorithmic HBM bytes of ONE instance's append (+ control query) pass at N live points: the packed factor
    N(N+1)/2, the whitened targets and inputs N n each, the UH B rows N (1+m) -- read once -- and what the append writes:
    the new factor row (N + 1), one row of Vw / X / UH B."""
    return itemsize * (N * (N + 1) // 2 + 2 * N * n + N * (1 + m) + (N + 1) + 2 * n + (1 + m))


def online_gp_growth(Bt, N0=128, N1=2048, dtype=torch.float64, device="cuda", seed=5, with_control=True, check=True,
                     reserved=True, fused=True, window=None, tail=False):
    """BASELINE configs[4]: every instance starts from an N0-point GP and takes one observation per control step
    until it holds N1 points -- the reference refits from scratch every `train_every_n_steps`
    (unicycle_move_to_pose.py:340-386).  reserved=True (default): capacity-reserving storage (`ops.ReservedGP`,
    capacity N1): an observation enters IN PLACE -- one streaming forward solve + O(N) bytes written, no allocation, no
    copies, no re-packing; the control step's posterior reads the same storage -- fused=True (default): on the SAME pass
    over the factors as the append's forward solve (`append(..., query=x)`; per segment `append_ms` is then that one pass
    + the in-place row writes, `control_step_ms` the solve launch alone).  reserved=False: `ops.gp_append` on the
    packed layout of exactly N points (every per-instance array copied per append, the operator re-packed every 32).
    window = W (reserved storage only): a sliding window over the most recent points (`ops.ReservedGP(window=W)`): the GP
    grows from N0 to W, then every 32nd append drops the oldest 32 points and refits the window (the drop is inside that
    step's `append_ms`); N1 is then the number of observations seen, the final check is against a from-scratch refit of the
    LAST window.
    tail=True (reserved storage, no window; N0 a multiple of 32): the appends since the last commit are contiguous rows beside the
    operator (`bcbf_gp_tail_step`), committed to the column layout 32 rows at a time (`bcbf_gp_tail_commit`: full-line writes) --
    the in-place append's one element per 128-byte line and step is what the next streaming pass waited for.
    Returns per-octave timings (HIP events) and the deviation of the final posterior from a from-scratch refit of all
    N1 points."""
    from .synthetic import make_instances, make_unicycle_task
    dev = torch.device(device)
    n, m = 3, 2
    p = make_instances(Bt, N1, n, m, dtype=dtype, device=dev, seed=seed)
    task = make_unicycle_task(Bt, dtype=dtype, device=dev, seed=seed + 1)
    cut = lambda t, N: t[:, :N].contiguous()
    Lop, UHB, info, _ = ops.refit(cut(p["X"], N0), cut(p["UH"], N0), p["Bm"], p["ell"], p["s2"], cut(p["jitter"], N0))
    assert int((info != 0).sum()) == 0
    Vw, _ = ops.potrs(Lop, cut(p["Xdot"], N0), cut(p["UH"], N0), p["M0"], want_alpha=False)
    X = cut(p["X"], N0)
    A = (0.01 * p["A"]).contiguous()
    ws = ops.control_workspace(Bt, 2, dtype, dev)
    x = task["x"].clone()
    if window is not None:
        assert reserved and N0 <= window, "a sliding window runs on reserved storage, from at most `window` points"
        rgp = ops.ReservedGP(Lop, Vw, X, UHB, p["ell"], p["s2"], p["Bm"], p["M0"], window + 32, window=window,
                             UH=cut(p["UH"], N0), Xdot=cut(p["Xdot"], N0), jitter=cut(p["jitter"], N0))
    else:
        rgp = ops.ReservedGP(Lop, Vw, X, UHB, p["ell"], p["s2"], p["Bm"], p["M0"], N1, tail=tail) if reserved else None
    if reserved:
        del Lop, Vw, UHB
    # pre-slice the observation stream (contiguous [N1][Bt,.]) so the timed loop holds only the path's own launches
    obs = [t.transpose(0, 1).contiguous() for t in (p["X"], p["UH"], p["Xdot"], p["jitter"])]
    edges = sorted({N0, N1} | {k for k in (256, 512, 1024, 2048) if N0 < k < N1} | ({window} if window and N0 < window < N1 else set()))
    segs = []
    fails = torch.zeros((), dtype=torch.int64, device=dev)
    for lo, hi in zip(edges[:-1], edges[1:]):
        k = hi - lo
        e = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(k)]
        torch.cuda.synchronize()
        for N in range(lo, hi):
            ev = e[N - lo]
            ev[0].record()
            if with_control and reserved and fused:
                # ONE pass over every instance's factor answers the control step's posterior query (on the N points) and the
                # forward solve of the append; (M_k, B_k) are then INPUTS of the fused task-rows / terms / SOCP launch
                info, _, _ = rgp.append(obs[0][N], obs[1][N], obs[2][N], obs[3][N], query=x, out=(ws["Mk"], ws["Bk"]))
                ev[1].record()
                ops.unicycle_control_step(dict(A=A), task, ws, x, dt=0.0, L_mean=4.0, max_iters=20)
                ev[2].record()
                fails += (info != 0).sum()
                continue
            if with_control and reserved:
                rgp.posterior(x, out=(ws["Mk"], ws["Bk"]))
                ops.unicycle_control_step(dict(A=A), task, ws, x, dt=0.0, L_mean=4.0, max_iters=20)
            elif with_control:
                gp = dict(Lop=Lop, Vw=Vw, X=X, UHB=UHB, ell=p["ell"], s2=p["s2"], Bm=p["Bm"], M0=p["M0"], A=A)
                ops.unicycle_control_step(gp, task, ws, x, dt=0.0, L_mean=4.0, max_iters=20)
            ev[1].record()
            if reserved:
                info = rgp.append(obs[0][N], obs[1][N], obs[2][N], obs[3][N])
            else:
                Lop, Vw, X, UHB, info = ops.gp_append(Lop, Vw, X, UHB, p["ell"], p["s2"], p["Bm"], p["M0"], obs[0][N],
                                                      obs[1][N], obs[2][N], obs[3][N])
            ev[2].record()
            fails += (info != 0).sum()             # stays on the device: the loop never waits for the host
        torch.cuda.synchronize()
        t_step = sum(ev[0].elapsed_time(ev[1]) for ev in e)
        t_app = sum(ev[1].elapsed_time(ev[2]) for ev in e)
        isz = p["X"].element_size()
        if with_control and reserved and fused:
            t_step, t_app = t_app, t_step              # (events: [0,1] = posterior + append, [1,2] = solve)
        seg = dict(N_from=lo, N_to=hi, control_step_ms=t_step / k, append_ms=t_app / k, step_ms=(t_step + t_app) / k,
                   append_GBs_algorithmic=Bt * isz * ((lo + hi) / 2) ** 2 / 2 / (t_app / k * 1e-3) / 1e9)
        if reserved and window is None:
            # roofline of the pass that dominates an append (+ the control query riding on it): every instance's packed factor,
            # whitened targets, inputs and UH B rows read once (SURVEY 8d's per-instance figure at the live N), summed over the
            # segment's appends; the O(N) bytes an append writes are counted too
            byt = sum(online_pass_bytes(N, n, m, isz) for N in range(lo, hi)) * Bt
            passes = 1 if (with_control and fused) or not with_control else 2
            gbs = byt * passes / (t_app * 1e-3) / 1e9 if passes == 1 else None
            seg["roofline"] = dict(bound="hbm", kernel="posterior_step_kernel<%s, %d, 4, 0, 1, false, 1> (query columns + the append's column on "
                                   "one pass) + %s" % ("double" if isz == 8 else "float", 1 + m,
                                                       "gp_tail_step_kernel (+ gp_tail_commit_kernel every 32nd step)" if tail else "gp_append_rows"),
                                   algorithmic_bytes_per_launch=byt / k, achieved=gbs, peak=8000.0, unit="GB/s",
                                   frac=None if gbs is None else gbs / 8000.0, traffic=None,
                                   how="sum over the segment's appends of Bt x online_pass_bytes(N) / sum of the HIP-event "
                                       "intervals around append(+query) [ms = append_ms]")
        segs.append(seg)
    out = dict(batch=Bt, N0=N0, N1=N1, dtype=str(dtype),
               storage=("reserved (" + ("row-major tail, committed 32 rows at a time" if tail else "in place") + ")"
                        + (", posterior query and append on one pass" if (fused and with_control) else ""))
               if reserved else "packed (copy per append)",
               segments=segs, append_failures=int(fails))
    if window is not None:
        out.update(window=window, drops=rgp.drops, live_points=rgp.N)
    if check:
        if window is not None:                           # the last window: points N1 - live .. N1 - 1 of the stream
            lo = N1 - rgp.N
            p = {k: (v[:, lo:N1].contiguous() if k in ("X", "UH", "Xdot", "jitter") else v) for k, v in p.items()}
        Lr, UHBr, info, _ = ops.refit(p["X"], p["UH"], p["Bm"], p["ell"], p["s2"], p["jitter"])
        Vr, _ = ops.potrs(Lr, p["Xdot"], p["UH"], p["M0"], want_alpha=False)
        if reserved:
            Mk, Bk = rgp.posterior(p["xq"])
        else:
            Mk, Bk = ops.posterior_step(Lop, Vw, X, UHB, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"])
        Mr, Br = ops.posterior_step(Lr, Vr, p["X"], UHBr, p["ell"], p["s2"], p["Bm"], p["M0"], p["xq"])
        prior = float((p["s2"][:, None, None] * p["Bm"]).abs().max())
        out["refit_failures"] = int((info != 0).sum())
        out["final_vs_refit"] = dict(Mk=float((Mk - Mr).abs().max() / max(1.0, float(Mr.abs().max()))),
                                     Bk=float((Bk - Br).abs().max() / prior))
    return out


class _PartsReserved:
    """The part batches' `ops.ReservedGP` objects of `learning_closed_loop(parts > 1)` seen as one (what the parity checks read)."""

    def __init__(self, rgps, bounds):
        self.rgps, self.bounds = rgps, bounds

    N = property(lambda self: self.rgps[0].N)
    tail = property(lambda self: self.rgps[0].tail)
    drops = property(lambda self: self.rgps[0].drops)
    drop_failures = property(lambda self: sum(g.drop_failures for g in self.rgps))
    _rJ = property(lambda self: torch.cat([g._rJ for g in self.rgps], 0))

    def posterior(self, xq):
        out = [g.posterior(xq[lo:hi].contiguous()) for g, (lo, hi) in zip(self.rgps, self.bounds)]
        return torch.cat([o[0] for o in out], 0), torch.cat([o[1] for o in out], 0)


def learning_closed_loop(Bt=4096, max_train=512, steps=200, refit_every=40, warmup=40, dtype=torch.float32, device="cuda",
                         seed=1234, schedule="online", n=3, m=2, barrier=None, parts=1, mid_period_steps=0):
    """The reference's REAL workload at BASELINE configs[2] scale: a control loop that keeps learning
    (`LearnedShiftInvariantDynamics.train`, unicycle_move_to_pose.py:340-386: buffer (x, u) every step, refit every
    `train_every_n_steps` = 40 on at most `max_train` points) -- Bt independent instances, each with its own GP over the most
    recent observations, never more than `max_train` of them.

    schedule = "online" (default): every step ONE pass over every instance's factor answers the control step's posterior
        query AND the forward solve of the new observation's in-place append (`ReservedGP.append(query=...)`), then the fused
        task rows / terms / SOCP / plant-step launch; when the model holds `max_train` points the oldest `refit_every` leave
        and the remaining window is refactored from the data (`ReservedGP(window=max_train - refit_every, drop=refit_every)`:
        the live size runs from max_train - refit_every to max_train - 1, so the padded size -- and with it the workgroup
        shape of the streaming kernel -- never exceeds max_train's): the model the controller queries is never more than zero
        steps old.
    schedule = "online_tail": the same schedule with the points observed since the last window refit kept as contiguous ROWS beside
        the window's factor (`ReservedGP(tail=True)`, `bcbf_gp_tail_step`): the streaming pass runs over the static window, a small
        tail kernel finishes the posterior over the newer points and appends the observation as one contiguous row -- no
        element-per-column writes into the operator, whose dirty lines cost the next pass a quarter of its time.
    schedule = "reference": the reference's cadence -- the GP is STATIC between refits (the headline control step,
        `bcbf_unicycle_control_step`: posterior pass + solve), observations only land in a buffer, every `refit_every`-th step
        the last `max_train` buffered points are refactored (`bcbf_refit` + `bcbf_potrs`).  `parts` > 1: the control steps
        run as `ops.ConcurrentControlLoop(parts=...)` (part batches on their own HIP streams, bench.py's default schedule: one
        part's latency-bound solve beside another part's HBM-bound posterior); a refit waits for all part streams (one host
        synchronisation per refit) and the part streams wait for it.

    Observations are PRE-DRAWN synthetic rows (`synthetic.make_instances`: well-conditioned random inputs) -- the cost of the
    learning loop on data that does not depend on the loop; `self_learning_closed_loop` below is the loop that learns from its own
    (x_t, u_t, x_{t+1}).  mid_period_steps: untimed extra steps after the timed region (fewer than refit_every), so that the final
    model the parity checks look at is MID-PERIOD -- a window plus appended / tail rows, not a model that was just refitted.

    `warmup` untimed steps (rounded up to whole refit periods so that the timed region starts right after a refit), then
    `steps` timed steps (a multiple of refit_every: every timed period holds exactly one refit) between two device
    synchronisations.  Returns the timings (wall clock for the total; HIP events for the shares), a roofline entry per
    kernel, and the final state for the parity checks: `final` = dict(rgp | gp tensors, raw window rows, query states)."""
    import time
    from .synthetic import make_instances, make_unicycle_task
    dev = torch.device(device)
    if steps % refit_every or steps <= 0:
        raise ValueError("steps must be a positive multiple of refit_every")
    warmup = -(-warmup // refit_every) * refit_every
    if not 0 <= mid_period_steps < refit_every:
        raise ValueError("mid_period_steps must be in [0, refit_every)")
    total = warmup + steps + int(mid_period_steps)
    window = max_train - refit_every if schedule in ("online", "online_tail") else max_train      # points the model holds right after a refit
    if window < 1:
        raise ValueError("max_train must exceed refit_every")
    p = make_instances(Bt, window + total, n, m, dtype=dtype, device=dev, seed=seed)
    task = make_unicycle_task(Bt, dtype=dtype, device=dev, seed=seed + 99)
    cut = lambda t, N: t[:, :N].contiguous()
    jit0 = cut(p["jitter"], window)
    for attempt in range(4):                                   # make_psd's retry (control_affine_model.py:899-921)
        Lop, UHB, info, _ = ops.refit(cut(p["X"], window), cut(p["UH"], window), p["Bm"], p["ell"], p["s2"], jit0)
        bad = info != 0
        if not bool(bad.any()):
            break
        jit0 = torch.where(bad[:, None], jit0 * 10, jit0).contiguous()
    assert int((info != 0).sum()) == 0, "initial refit failed"
    Vw, _ = ops.potrs(Lop, cut(p["Xdot"], window), cut(p["UH"], window), p["M0"], want_alpha=False)
    isz = p["X"].element_size()
    A = p["A"]
    ws = ops.control_workspace(Bt, 2, dtype, dev)
    x = task["x"].clone()
    dt_plant, L_true, L_mean = 1e-3, 1.0, 4.0
    obs = [t.transpose(0, 1).contiguous() for t in (p["X"], p["UH"], p["Xdot"], p["jitter"])]     # [N][Bt, .]
    fails = torch.zeros((), dtype=torch.int64, device=dev)
    fails_vec = torch.zeros(Bt, dtype=torch.int32, device=dev)
    online = schedule in ("online", "online_tail")
    if online and parts > 1:
        # part batches on their own streams (instances never interact): one part's latency-bound solve and its small tail / row
        # kernels run beside another part's streaming pass, as ops.ConcurrentControlLoop does for the static model; a part's window
        # refit runs on its own stream too (its jitter-retry check waits for that stream only)
        base, rem = divmod(Bt, parts)
        bounds = [(c * base + min(c, rem), (c + 1) * base + min(c + 1, rem)) for c in range(parts)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
        cur = torch.cuda.current_stream(dev)
        X0, UH0, Y0 = cut(p["X"], window), cut(p["UH"], window), cut(p["Xdot"], window)
        tkeys = ops.ConcurrentControlLoop.TASK_INSTANCE_KEYS
        rgps, solves = [], []
        for c in range(parts):
            sl = slice(*bounds[c])
            streams[c].wait_stream(cur)
            with torch.cuda.stream(streams[c]):
                rgps.append(ops.ReservedGP(Lop[sl], Vw[sl], X0[sl], UHB[sl], p["ell"][sl], p["s2"][sl], p["Bm"][sl], p["M0"][sl],
                                           window + refit_every, window=window, drop=refit_every, UH=UH0[sl], Xdot=Y0[sl],
                                           jitter=jit0[sl], tail=schedule == "online_tail"))
            taskc = {k: (v[sl] if (torch.is_tensor(v) and k in tkeys) else v) for k, v in task.items()}
            Ac = A[sl] if (A.dim() == 3 and A.shape[0] == Bt and Bt > 1) else A
            solves.append(ops.unicycle_control_step_prepare(dict(A=Ac), taskc, {k: v[sl] for k, v in ws.items()}, x[sl], dt=dt_plant,
                                                            L_true=L_true, L_mean=L_mean, clf_gamma=10.0, max_iters=20,
                                                            stream=streams[c]))
        for s_ in streams:
            s_.synchronize()
        del Lop, Vw, UHB, X0, UH0, Y0
        rgp = _PartsReserved(rgps, bounds)
        # per-part views, made once (the loop's host side is what bounds three and four part batches)
        obs_c = [[o[:, slice(*bounds[c])] for o in obs] for c in range(parts)]
        x_c = [x[slice(*bounds[c])] for c in range(parts)]
        out_c = [(ws["Mk"][slice(*bounds[c])], ws["Bk"][slice(*bounds[c])]) for c in range(parts)]
        fails_c = [fails_vec[slice(*bounds[c])] for c in range(parts)]
    elif online:
        rgp = ops.ReservedGP(Lop, Vw, cut(p["X"], window), UHB, p["ell"], p["s2"], p["Bm"], p["M0"], window + refit_every,
                             window=window, drop=refit_every, UH=cut(p["UH"], window), Xdot=cut(p["Xdot"], window), jitter=jit0,
                             tail=schedule == "online_tail")
        del Lop, Vw, UHB
        solve = ops.unicycle_control_step_prepare(dict(A=A), task, ws, x, dt=dt_plant, L_true=L_true, L_mean=L_mean,
                                                  clf_gamma=10.0, max_iters=20)
    elif schedule == "reference":
        gp = dict(Lop=Lop, Vw=Vw, X=cut(p["X"], window), UHB=UHB, ell=p["ell"], s2=p["s2"], Bm=p["Bm"], M0=p["M0"], A=A)
        jit_w = jit0
        lo = 0
        if parts > 1:
            loop = ops.ConcurrentControlLoop(gp, task, x, parts=parts, dt=dt_plant, L_true=L_true, L_mean=L_mean, clf_gamma=10.0,
                                             max_iters=20)
            ws = loop.ws
        else:
            step_fn = ops.unicycle_control_step_prepare(gp, task, ws, x, dt=dt_plant, L_true=L_true, L_mean=L_mean,
                                                        clf_gamma=10.0, max_iters=20)
    else:
        raise ValueError("schedule: 'online', 'online_tail' or 'reference'")
    E = lambda: torch.cuda.Event(enable_timing=True)
    ev = [[E(), E(), E(), E()] for _ in range(total)]          # step start / pass end / solve end / (refit end)
    for row in ev:                                             # (torch creates the hipEvent handle at the first record; the
        for e_ in row:                                         #  reference schedule hands raw handles to the C entry point)
            e_.record()
    concurrent = parts > 1
    evp = None
    if concurrent:                                             # per part: events around its posterior launch, on its stream
        if not online:
            streams = loop.streams
        evp = [[(E(), E()) for _ in range(parts)] for _ in range(total)]
        for row in evp:
            for c, (a_, b_) in enumerate(row):
                a_.record(streams[c]); b_.record(streams[c])
        ev_base = E()
    refit_steps = []
    t0 = elapsed = None
    for t in range(total):
        if t == warmup:
            if barrier is not None:
                barrier()
            torch.cuda.synchronize(dev)
            if concurrent:
                for s_ in streams:
                    s_.synchronize()
                ev_base.record(streams[0])
            t0 = time.perf_counter()
        if t == warmup + steps:                                  # (mid_period_steps > 0: the timed region ends here)
            torch.cuda.synchronize(dev)
            if barrier is not None:
                barrier()
            elapsed = time.perf_counter() - t0
        N_obs = window + t
        e = ev[t]
        e[0].record()
        if online and concurrent:
            drops_before = rgp.drops
            for c in range(parts):
                oc = obs_c[c]
                with torch.cuda.stream(streams[c]):
                    evp[t][c][0].record(streams[c])
                    info, _, _ = rgps[c].append(oc[0][N_obs], oc[1][N_obs], oc[2][N_obs], oc[3][N_obs], query=x_c[c], out=out_c[c])
                    evp[t][c][1].record(streams[c])
                    solves[c]()
                    fails_c[c] += info != 0
            if rgp.drops != drops_before:
                refit_steps.append(t)
        elif online:
            drops_before = rgp.drops
            info, _, _ = rgp.append(obs[0][N_obs], obs[1][N_obs], obs[2][N_obs], obs[3][N_obs], query=x, out=(ws["Mk"], ws["Bk"]))
            # (the window's drop + refit, when this append filled it, ran inside append -- timed below as its own share)
            e[1].record()
            solve()
            e[2].record()
            fails_vec += info != 0                       # (stays on the device: the loop never waits for the host)
            if rgp.drops != drops_before:
                refit_steps.append(t)
        else:
            if concurrent:
                loop.step(evp[t])
            else:
                step_fn(e[0], e[1])                              # (events around the posterior launch, on its stream)
                e[2].record()
            if (t + 1) % refit_every == 0:
                if concurrent:
                    loop.synchronize()                           # the refit overwrites what the part streams read
                    e[2].record()
                lo = t + 1
                sl = slice(lo, lo + window)
                Xw, UHw, Yw = (p[k][:, sl].contiguous() for k in ("X", "UH", "Xdot"))
                jit_w = p["jitter"][:, sl].contiguous()
                for attempt in range(4):
                    ops.refit(Xw, UHw, p["Bm"], p["ell"], p["s2"], jit_w, out=(gp["Lop"], gp["UHB"], info))
                    bad = info != 0
                    if not bool(bad.any()):
                        break
                    jit_w = torch.where(bad[:, None], jit_w * 10, jit_w).contiguous()
                ops.potrs(gp["Lop"], Yw, UHw, p["M0"], want_alpha=False, out_Vw=gp["Vw"])
                gp["X"].copy_(Xw)
                fails += (info != 0).sum()
                e[3].record()
                if concurrent:
                    for st_ in loop.streams:
                        st_.wait_stream(torch.cuda.current_stream(dev))
                refit_steps.append(t)
    torch.cuda.synchronize(dev)
    if elapsed is None:
        if barrier is not None:
            barrier()
        elapsed = time.perf_counter() - t0
    timed = range(warmup, warmup + steps)
    if online and concurrent:
        # the parts drift apart, so shares are taken over the whole timed region: the time during which at least one part's pass
        # (on a refit step: pass + window refit) ran, and a refit's own share from each part's refit-step interval minus that part's
        # mean plain interval
        spans = sorted((ev_base.elapsed_time(a_), ev_base.elapsed_time(b_)) for t in timed for a_, b_ in evp[t])
        busy, ca, cb = 0.0, spans[0][0], spans[0][1]
        for a_, b_ in spans[1:]:
            if a_ > cb:
                busy += cb - ca
                ca, cb = a_, b_
            else:
                cb = max(cb, b_)
        busy += cb - ca
        tr = set(refit_steps)
        dur = lambda t, c: evp[t][c][0].elapsed_time(evp[t][c][1])
        plain_c = [sum(dur(t, c) for t in timed if t not in tr) / max(1, sum(1 for t in timed if t not in tr)) for c in range(parts)]
        rs = [t for t in timed if t in tr]
        n_refits = len(rs)
        refit_ms = (sum(dur(t, c) - plain_c[c] for t in rs for c in range(parts)) / (n_refits * parts)) if rs else 0.0
        pass_ms = busy / steps - refit_ms * n_refits / steps / parts      # (a part's refit occupies the device for ~1/parts of the batch)
    elif online:
        # the append (+ the drop/refit on refit steps) sits between e[0] and e[1]; a refit step's own share = its interval minus
        # the mean interval of the other steps at a comparable N
        tr = set(refit_steps)
        plain = [ev[t][0].elapsed_time(ev[t][1]) for t in timed if t not in tr]
        withr = [ev[t][0].elapsed_time(ev[t][1]) for t in timed if t in tr]
        pass_ms = sum(plain) / max(1, len(plain))
        refit_ms = (sum(withr) / max(1, len(withr)) - pass_ms) if withr else 0.0
        n_refits = len(withr)
    elif concurrent:
        # the part batches' posterior launches overlap one another: the time during which at least one of them ran (bench.py)
        spans = sorted((ev_base.elapsed_time(a_), ev_base.elapsed_time(b_)) for t in timed for a_, b_ in evp[t])
        busy, ca, cb = 0.0, spans[0][0], spans[0][1]
        for a_, b_ in spans[1:]:
            if a_ > cb:
                busy += cb - ca
                ca, cb = a_, b_
            else:
                cb = max(cb, b_)
        pass_ms = (busy + cb - ca) / steps
        rs = [t for t in refit_steps if warmup <= t < warmup + steps]
        refit_ms = sum(ev[t][2].elapsed_time(ev[t][3]) for t in rs) / max(1, len(rs))
        n_refits = len(rs)
    else:
        pass_ms = sum(ev[t][0].elapsed_time(ev[t][1]) for t in timed) / steps
        rs = [t for t in refit_steps if warmup <= t < warmup + steps]
        refit_ms = sum(ev[t][2].elapsed_time(ev[t][3]) for t in rs) / max(1, len(rs))
        n_refits = len(rs)
    if os.environ.get("BCBF_LEARN_DUMP"):          # development: per-step pass / solve intervals with the live size
        import json as _json
        _json.dump([dict(t=t, N=(window + (t % refit_every)) if online else window, pass_ms=ev[t][0].elapsed_time(ev[t][1]),
                         solve_ms=ev[t][1].elapsed_time(ev[t][2])) for t in timed], open(os.environ["BCBF_LEARN_DUMP"], "w"))
    solve_ms = 0.0 if concurrent else sum(ev[t][1].elapsed_time(ev[t][2]) for t in timed) / steps     # (concurrent: hidden beside the passes)
    ms_step = elapsed / steps * 1e3
    # roofline entries.  pass: every instance's packed factor + whitened targets + inputs + UH B rows read once at the live N
    if online:
        live = [window + (t % refit_every) for t in timed if t not in set(refit_steps)]
        pass_bytes = sum(online_pass_bytes(N, n, m, isz) for N in live) / max(1, len(live)) * Bt
        pass_kernel = "posterior_step_kernel<%s, %d, 4, 0, 1, false, 1> + %s" % ("float" if isz == 4 else "double", 1 + m,
                                                                                   "gp_tail_step_kernel" if rgp.tail else "gp_append_inplace_kernel")
    else:
        pass_bytes = isz * (window * (window + 1) // 2 + 2 * window * n + window * (1 + m)) * Bt
        pass_kernel = "posterior_step_kernel<%s, %d, 4, 0, 1, false, 0>" % ("float" if isz == 4 else "double", 1 + m)
    peak_t = 157.3 if isz == 4 else 78.6
    refit_flops = (Bt / parts if (online and concurrent) else Bt) * window ** 3 / 3.0     # (online on part batches: a refit is one part's)
    roof = {"pass": dict(bound="hbm", kernel=pass_kernel, algorithmic_bytes_per_launch=pass_bytes,
                         achieved=pass_bytes / (pass_ms * 1e-3) / 1e9, peak=8000.0, unit="GB/s",
                         frac=pass_bytes / (pass_ms * 1e-3) / 1e9 / 8000.0, kernel_ms=pass_ms, traffic=None),
            "refit": dict(bound="mfma", kernel="refit_wave_kernel<%s, ...> (+ bcbf_potrs%s)" % ("float" if isz == 4 else "double",
                                                                                                ", row moves, bcbf_gp_reserve" if online else ""),
                          algorithmic_flops_per_launch=refit_flops, achieved=refit_flops / (refit_ms * 1e-3) / 1e12 if refit_ms > 0 else None,
                          peak=peak_t, unit="TFLOP/s", frac=refit_flops / (refit_ms * 1e-3) / 1e12 / peak_t if refit_ms > 0 else None,
                          kernel_ms=refit_ms, traffic=None,
                          note="over the WHOLE refit share of a refit step (every launch of it), a lower bound on the kernel's own rate")}
    out = dict(schedule=schedule, batch=Bt, max_train=max_train, points_after_refit=window, steps=steps, warmup=warmup, refit_every=refit_every, dtype=str(dtype),
               seconds=elapsed, ms_per_step=ms_step, instance_steps_per_s=Bt * steps / elapsed,
               shares=dict(pass_ms_per_step=pass_ms, solve_ms_per_step=solve_ms, refit_ms_per_refit=refit_ms,
                           refit_ms_per_step=refit_ms * n_refits / steps, refits_in_timed_region=n_refits,
                           other_ms_per_step=ms_step - pass_ms - solve_ms - refit_ms * n_refits / steps),
               roofline=roof, append_or_refit_failures=int(fails) + int((fails_vec != 0).sum()), parts=parts, solver_optimal_fraction=float((ws["status"] == 0).float().mean()))
    if online:
        out["drop_failures"] = rgp.drop_failures
        lo = window + total - rgp.N
        final = dict(rgp=rgp, lo=lo, N=rgp.N, jitter=rgp._rJ[:, :rgp.N])
    else:
        final = dict(gp=gp, lo=lo, N=window, jitter=jit_w)
    final.update(p=p, x=x, ws=ws)
    return out, final


def final_window_vs_device_refit(final, sample=64):
    """Self-check of a learning loop's final model (any dtype) against a from-scratch fp64 refit ON THE DEVICE of the same
    window rows with the jitter every point ended up with: max deviation of the posterior at the instances' query points over
    `sample` instances spread over the batch, relative to max(1, |M_k|) and to the prior scale.  (The parity check against
    the CPU oracle lives in tests/.)"""
    p, lo, N = final["p"], final["lo"], final["N"]
    Bt = p["X"].shape[0]
    idx = torch.linspace(0, Bt - 1, min(sample, Bt), device=p["X"].device).long()
    f64 = lambda t: t.double().contiguous()
    sel = lambda k: f64(p[k][idx, lo:lo + N])
    hp = {k: f64(p[k][idx]) for k in ("Bm", "ell", "s2", "M0", "xq")}
    Lr, UHBr, info, _ = ops.refit(sel("X"), sel("UH"), hp["Bm"], hp["ell"], hp["s2"], f64(final["jitter"][idx]))
    Vr, _ = ops.potrs(Lr, sel("Xdot"), sel("UH"), hp["M0"], want_alpha=False)
    Mr, Br = ops.posterior_step(Lr, Vr, sel("X"), UHBr, hp["ell"], hp["s2"], hp["Bm"], hp["M0"], hp["xq"])
    if "rgp" in final:
        Mk, Bk = final["rgp"].posterior(p["xq"])
    else:
        g = final["gp"]
        Mk, Bk = ops.posterior_step(g["Lop"], g["Vw"], g["X"], g["UHB"], g["ell"], g["s2"], g["Bm"], g["M0"], p["xq"])
    prior = (hp["s2"][:, None, None] * hp["Bm"]).abs().amax(dim=(1, 2))
    return dict(instances=int(idx.numel()), refit_failures=int((info != 0).sum()),
                Mk=float(((f64(Mk[idx]) - Mr).abs().amax(dim=(1, 2)) / Mr.abs().amax(dim=(1, 2)).clamp(min=1.0)).max()),
                Bk=float(((f64(Bk[idx]) - Br).abs().amax(dim=(1, 2)) / prior).max()))


def self_learning_closed_loop(Bt=4096, max_train=512, steps=200, refit_every=40, warmup=None, dtype=torch.float32, device="cuda",
                              seed=1234, schedule="reference", parts=4, stagger=True, shift_invariant=True, dt=0.01,
                              retry_levels=3, fit_iters=0, fit_lr=0.1, fit_dtype=torch.float64, record_states=False, barrier=None,
                              mid_period_steps=0, query_shift_invariant=True, level_decay_every=4, factor_dtype=None,
                              min_jitter_level=1e-5):
    """The reference's learning loop for MANY instances, fed BY ITSELF (LearnedShiftInvariantDynamics.train / fit,
    unicycle_move_to_pose.py:326-386): every control step's observation row is built on the device from the loop's own
    (x_t, u_t, x_{t+1}) -- inside the solve / plant launch (`bcbf_unicycle_control_step_observe`): regressor input = the state
    before the step, shift invariant (0, 0, theta) (:326-330), target = (x_{t+1} - x_t) / dt minus the mean model
    `AckermannDrive(L_mean)` (:364-372) -- and is what the next refit / append learns from.  query_shift_invariant (default):
    the controller queries the learned model at the shift-invariant input of the current state too, which the kernel keeps in
    `xq` (xq_next).  The reference's `fu_func_gp` (:388-397) does NOT go through the wrapper -- its controller queries a model
    trained on (0, 0, theta) at the raw (x, y, theta), far from every training input, so the posterior is the prior and most
    chance constraints are infeasible (measured here: 13 % of the programs solve, against 87 % with consistent inputs);
    query_shift_invariant=False reproduces that.
    Jitter (make_psd, :899-921): every factorisation draws level * rand per point, x10 per failed attempt; an instance starts a
    refit at the level that last worked and goes one level down every `level_decay_every`-th refit (the reference restarts at
    1e-5 every time; in fp32 on a trajectory's near-collinear rows that level fails for three instances in four, each failure
    a whole factorisation).

    Bt instances in `parts` part batches, each on its own HIP stream with its own model buffers; NOTHING waits for the host:
    a refit is `bcbf_refit` + `retry_levels` unconditional `bcbf_refit_retry` launches (make_psd's x10 schedule on the failed
    instances) + `bcbf_potrs` into the operator buffer that is not being read, then the buffers swap.  `stagger`: part c refits
    at steps = c * refit_every / parts (mod refit_every), so one part's matrix-core refit runs beside the other parts' HBM-bound
    passes instead of stopping the device.
      schedule "reference":   the model is static between refits (posterior pass + solve); every `refit_every` steps the last
                              `max_train` observations are refactored -- the reference's cadence (train_every_n_steps).
      schedule "online_tail": every observation enters the model the step after it was made (`ReservedGP(tail=True)`:
                              streaming pass over the window + tail kernel), window refit every `refit_every` appends.
    The windows start filled with synthetic rows (the model the run starts from); `warmup` (default: enough periods to flush
    them, >= window + refit_every steps) makes the timed region learn from the loop's own rows only.
    fit_iters > 0 (schedule "reference"): every refit first runs `fit_iters` Adam iterations of the marginal likelihood on the
    part's window for every instance (`BatchedHyperFit`, in `fit_dtype`) -- the reference's `fit(..., training_iter=100)`;
    the hyper-parameters the control path reads are updated in place.
    mid_period_steps: untimed extra steps after the timed region, so that the final model is mid-period (tail / window not
    just refitted) for the parity checks.
    factor_dtype = torch.float64 with dtype = torch.float32 (schedule "reference"): MIXED precision -- the window is factored in
    fp64 (`bcbf_refit` + `bcbf_potrs` on the rows cast up: cond(K_b) ~ N s2 / jitter is beyond fp32 on a trajectory's rows, the
    fp64 factorisation succeeds at make_psd's base level with no retry) and the operator, `UH B`, `Vw` are ROUNDED to fp32 for
    the streaming passes, which are HBM bound and move half the bytes.  The error the passes then add is cond(L) eps32 =
    sqrt(cond K_b) eps32 ~ 1e-4, not cond(K_b) eps32: the loop runs at fp32's pass rate with models that agree with the fp64
    refit of their rows to ~1e-3 (reported: final_vs_fp64_refit_on_device).
    min_jitter_level (default make_psd's 1e-5): the floor of every instance's jitter level.  fp32 PASSES cannot resolve a posterior
    variance below ~sqrt(cond K_b) eps32 of the prior: with fp64 factors at the 1e-5 level B_k = s2 B - W'W is rounding noise (it
    goes indefinite and the cone conversion refuses the program); 1e-3 keeps it resolvable.

    Returns (report, final): final = dict(rows = the raw observation rows each instance's model holds, oldest first
    (X, UH, Y, jitter [Bt, N, .]), posterior = (Mk, Bk) of the final model at `xq_check`, xq_check, hyper-parameters; states
    (record_states): the visited (x_t, u_t) of every step)."""
    import time
    from .synthetic import make_instances, make_unicycle_task
    dev = torch.device(device)
    n, m = 3, 2
    if schedule not in ("reference", "online_tail"):
        raise ValueError("schedule: 'reference' or 'online_tail'")
    if steps % refit_every or steps <= 0:
        raise ValueError("steps must be a positive multiple of refit_every")
    if fit_iters and schedule != "reference":
        raise ValueError("fit_iters: the hyper-parameter fit rides on the reference schedule's refits")
    online = schedule == "online_tail"
    window = max_train - refit_every if online else max_train
    if window < 1:
        raise ValueError("max_train must exceed refit_every")
    if warmup is None:
        warmup = window + refit_every
    warmup = -(-warmup // refit_every) * refit_every
    total = warmup + steps + int(mid_period_steps)
    Ntot = window + total + 1
    p = make_instances(Bt, window, n, m, dtype=dtype, device=dev, seed=seed, variant="theta" if shift_invariant else "dense")
    task = make_unicycle_task(Bt, dtype=dtype, device=dev, seed=seed + 99)
    f = dict(dtype=dtype, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 7)
    # the observation stream of every instance: rows 0 .. window-1 = the synthetic start, row window + t = step t's observation
    Xall, UHall, Yall = torch.zeros(Bt, Ntot, n, **f), torch.zeros(Bt, Ntot, 1 + m, **f), torch.zeros(Bt, Ntot, n, **f)
    # make_psd's draws (:907-910): reference schedule -- the jitter every row was last factored with (filled in per refit); online --
    # one rand per step, scaled by the instance's level when the point enters
    Jall = torch.rand(Bt, Ntot, generator=gen, **f).contiguous()
    Jall[:, :window] = p["jitter"]
    Xall[:, :window], UHall[:, :window], Yall[:, :window] = p["X"], p["UH"], p["Xdot"]
    x = task["x"].clone()
    # the planner's target moves along the straight line start -> goal at constant speed and reaches the goal when the run ends
    # (PiecewiseLinearPlanner, planner.py:54-64); the solve / plant launch advances it (flags bit 1)
    task["dot_plan"] = ((task["xg"] - task["x"]) / (total * dt)).contiguous()
    task["plan"] = (task["x"] + 20 * dt * task["dot_plan"]).contiguous()        # (a look-ahead: at the state itself the CLF's polar terms are singular)
    L_true, L_mean = 1.0, 4.0
    base, rem = divmod(Bt, parts)
    bounds = [(c * base + min(c, rem), (c + 1) * base + min(c + 1, rem)) for c in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    cur = torch.cuda.current_stream(dev)
    ws = ops.control_workspace(Bt, 2, dtype, dev)
    rnd = lambda *shape: torch.rand(*shape, generator=gen, **f)
    tkeys = ops.ConcurrentControlLoop.TASK_INSTANCE_KEYS
    offsets = [(c * refit_every) // parts if stagger else 0 for c in range(parts)]
    hyper = {k: p[k] for k in ("ell", "s2", "Bm", "M0", "A")}                       # updated in place by the fit
    xs_log, us_log = ([], []) if record_states else (None, None)

    class Part:
        pass
    P = []
    for c in range(parts):
        lo, hi = bounds[c]
        sl = slice(lo, hi)
        pt = Part()
        pt.sl, pt.Bt = sl, hi - lo
        pt.X, pt.UH, pt.Y, pt.J = Xall[sl], UHall[sl], Yall[sl], Jall[sl]             # contiguous [Bt_c, Ntot, .] views
        pt.hp = {k: v[sl] for k, v in hyper.items()}
        pt.x = x[sl]
        pt.n_refits = 0
        # the learned model's query: the shift-invariant input of the current state, kept by the solve / plant launch (xq_next)
        pt.xq = None
        if query_shift_invariant and shift_invariant:
            pt.xq = pt.x.clone()
            pt.xq[:, :2] = 0
        obs_kw = dict(xq=pt.xq, xq_next=pt.xq, shift_invariant=shift_invariant, advance_plan=True)
        taskc = {k: (v[sl] if (torch.is_tensor(v) and k in tkeys) else v) for k, v in task.items()}
        wsc = {k: v[sl] for k, v in ws.items()}
        pt.ws = wsc
        streams[c].wait_stream(cur)
        with torch.cuda.stream(streams[c]):
            cutw = lambda t_: t_[:, :window].contiguous()
            Xw, UHw, Yw, Jw = cutw(pt.X), cutw(pt.UH), cutw(pt.Y), cutw(pt.J)
            E = ops.lop_elems(window, dtype)
            mk = lambda: dict(Lop=torch.empty(pt.Bt, E, **f), UHB=torch.empty(pt.Bt, window, 1 + m, **f), Vw=torch.empty(pt.Bt, window, n, **f),
                              X=torch.empty(pt.Bt, window, n, **f))
            pt.info, pt.info2 = torch.zeros(pt.Bt, dtype=torch.int32, device=dev), torch.zeros(pt.Bt, dtype=torch.int32, device=dev)
            pt.fail_count = torch.zeros((), dtype=torch.int64, device=dev)
            pt.level = torch.full((pt.Bt,), float(min_jitter_level), **f)     # make_psd's level per instance (x10 per failed attempt)
            pt.retry_counts = torch.zeros(retry_levels + 1, dtype=torch.int64, device=dev)

            mixed = factor_dtype is not None and factor_dtype != dtype
            if mixed:
                fw = dict(dtype=factor_dtype, device=dev)
                pt.wide = dict(Lop=torch.empty(pt.Bt, ops.lop_elems(window, factor_dtype), **fw), UHB=torch.empty(pt.Bt, window, 1 + m, **fw),
                               Vw=torch.empty(pt.Bt, window, n, **fw))

            def factor_into(buf, Xw, UHw, Yw, Jw, pt=pt, mixed=mixed):
                if mixed:
                    # factor in the wide precision, round the results into the buffers the passes read (the packed layout is the
                    # same element for element in both precisions: a cast, no re-layout)
                    up = lambda t_: t_.to(factor_dtype)
                    hpw = {k: up(v) for k, v in pt.hp.items()}
                    Xd, UHd, Yd, Jd = up(Xw), up(UHw), up(Yw), up(Jw)
                    w = pt.wide
                    ops.refit_with_retries(Xd, UHd, hpw["Bm"], hpw["ell"], hpw["s2"], Jd, (w["Lop"], w["UHB"], pt.info),
                                           levels=retry_levels, scratch=pt.info2, level=pt.level, counts=pt.retry_counts)
                    ops.potrs(w["Lop"], Yd, UHd, hpw["M0"], want_alpha=False, out_Vw=w["Vw"])
                    buf["Lop"].copy_(w["Lop"]); buf["UHB"].copy_(w["UHB"]); buf["Vw"].copy_(w["Vw"])
                    Jw.copy_(Jd)
                else:
                    ops.refit_with_retries(Xw, UHw, pt.hp["Bm"], pt.hp["ell"], pt.hp["s2"], Jw, (buf["Lop"], buf["UHB"], pt.info),
                                           levels=retry_levels, scratch=pt.info2, level=pt.level, counts=pt.retry_counts)
                    ops.potrs(buf["Lop"], Yw, UHw, pt.hp["M0"], want_alpha=False, out_Vw=buf["Vw"])
                buf["X"].copy_(Xw)
                pt.fail_count += (pt.info != 0).sum()
            pt.factor_into = factor_into
            if online:
                b0 = mk()
                factor_into(b0, Xw, UHw, Yw, Jw)
                # stagger: part c starts `offset` rows into its period (N_init = window + offset): its drops come that much earlier
                pt.rgp = ops.ReservedGP(b0["Lop"], b0["Vw"], Xw, b0["UHB"], pt.hp["ell"], pt.hp["s2"], pt.hp["Bm"], pt.hp["M0"],
                                        window + refit_every, window=window, drop=refit_every, UH=UHw, Xdot=Yw, jitter=Jw, tail=True,
                                        retry_levels=retry_levels, factor_dtype=factor_dtype, min_jitter_level=min_jitter_level)
                pt.rgp.level_decay_every = level_decay_every
                pt.obs = [tuple(torch.zeros(pt.Bt, 3, **f) for _ in range(3)) for _ in range(2)]
                pt.solve = ops.unicycle_control_step_prepare(dict(A=pt.hp["A"]), taskc, wsc, pt.x, dt=dt, L_true=L_true, L_mean=L_mean,
                                                             clf_gamma=10.0, max_iters=20, stream=streams[c],
                                                             observe=obs_kw)
            else:
                pt.bufs = [mk(), mk()]
                factor_into(pt.bufs[0], Xw, UHw, Yw, Jw)
                pt.cur = 0
                pt.step = [ops.unicycle_control_step_prepare(dict(b_, **pt.hp), taskc, wsc, pt.x, dt=dt, L_true=L_true, L_mean=L_mean,
                                                             clf_gamma=10.0, max_iters=20, stream=streams[c],
                                                             observe=obs_kw)
                           for b_ in pt.bufs]
                pt.lo = 0
                if fit_iters:
                    from .batched_fit import BatchedHyperFit
                    pt.bf = BatchedHyperFit.from_values(pt.hp["A"], pt.hp["Bm"], pt.hp["ell"], pt.hp["s2"], pt.hp["M0"], dtype=fit_dtype)
        P.append(pt)
    for s_ in streams:
        s_.synchronize()
    if online:
        # bootstrap: the first append needs an observation -- one control step on the start model (posterior, solve + observe)
        for c, pt in enumerate(P):
            with torch.cuda.stream(streams[c]):
                pt.q = pt.xq if pt.xq is not None else pt.x
                pt.rgp.posterior(pt.q, out=(pt.ws["Mk"], pt.ws["Bk"]))
                pt.solve(obs=(*pt.obs[0], 1))
                # stagger the parts' drops: part c enters `offset` more observations before the loop (untimed)
                for k in range(offsets[c]):
                    o = pt.obs[k % 2]
                    pt.rgp.append(o[0], o[1], o[2], pt.rgp.jitter_level * pt.J[:, window + total - 1 - k], query=pt.q, out=(pt.ws["Mk"], pt.ws["Bk"]))
                    pt.solve(obs=(*pt.obs[(k + 1) % 2], 1))
                pt.k0 = offsets[c]
        for s_ in streams:
            s_.synchronize()
    E_ = lambda: torch.cuda.Event(enable_timing=True)
    evp = [[(E_(), E_()) for _ in range(parts)] for _ in range(total)]
    evr = {}
    for row in evp:
        for c, (a_, b_) in enumerate(row):
            a_.record(streams[c]); b_.record(streams[c])
    ev_base = E_()
    refit_log = []

    def do_refit(c, pt, t):
        """Part c, after step t: refactor the last `window` observations into the buffer that is not being read; swap."""
        lo = t + 1
        cutw = lambda t_: t_[:, lo:lo + window].contiguous()
        Xw, UHw, Yw = cutw(pt.X), cutw(pt.UH), cutw(pt.Y)
        # a fresh jitter draw per factorisation, at the instance's level: one below the one that last worked (make_psd, :903-919)
        pt.n_refits += 1
        if level_decay_every and pt.n_refits % level_decay_every == 0:
            pt.level.div_(10).clamp_(min=float(min_jitter_level))
        Jw = (pt.level[:, None] * rnd(pt.Bt, window)).contiguous()
        if fit_iters:
            wd = fit_dtype
            pt.bf.fit(Xw.to(wd), UHw[:, :, 1:].to(wd).contiguous(), Yw.to(wd), training_iter=fit_iters, lr=fit_lr)
            hp = pt.bf.derive()
            for k in ("ell", "s2", "Bm", "M0", "A"):
                pt.hp[k].copy_(hp[k].to(dtype))
        nxt = 1 - pt.cur
        pt.factor_into(pt.bufs[nxt], Xw, UHw, Yw, Jw)
        pt.J[:, lo:lo + window] = Jw                                  # the level every point was finally factored with
        pt.cur, pt.lo = nxt, lo

    t0 = None
    for t in range(total):
        if t == warmup:
            if barrier is not None:
                barrier()
            torch.cuda.synchronize(dev)
            ev_base.record(streams[0])
            t0 = time.perf_counter()
        if t == warmup + steps:
            torch.cuda.synchronize(dev)
            if barrier is not None:
                barrier()
            elapsed = time.perf_counter() - t0
        if record_states:
            torch.cuda.synchronize(dev)
            xs_log.append(x.clone())
        for c, pt in enumerate(P):
            row = window + t
            if online:
                with torch.cuda.stream(streams[c]):
                    k = pt.k0 + t
                    o_prev, o_next = pt.obs[k % 2], pt.obs[(k + 1) % 2]
                    drops_before = pt.rgp.drops
                    evp[t][c][0].record(streams[c])
                    jit = pt.rgp.jitter_level * pt.J[:, row]                  # (J holds rand draws here: the new point's jitter at the instance's level)
                    pt.rgp.append(o_prev[0], o_prev[1], o_prev[2], jit, query=pt.q, out=(pt.ws["Mk"], pt.ws["Bk"]))
                    evp[t][c][1].record(streams[c])
                    pt.solve(obs=(*o_next, 1))
                    if pt.rgp.drops != drops_before:
                        refit_log.append((t, c))
            else:
                pt.step[pt.cur](evp[t][c][0], evp[t][c][1], obs=(pt.X[:, row], pt.UH[:, row], pt.Y[:, row], Ntot))
                if (t + 1 - offsets[c]) % refit_every == 0 and t + 1 >= refit_every:
                    with torch.cuda.stream(streams[c]):
                        a_, b_ = E_(), E_()
                        a_.record(streams[c])
                        do_refit(c, pt, t)
                        b_.record(streams[c])
                        evr[(t, c)] = (a_, b_)
                    refit_log.append((t, c))
        if record_states:
            torch.cuda.synchronize(dev)
            # the control that was APPLIED: the program's solution, or zero for an instance whose program was not solved (frozen)
            us_log.append(torch.where((ws["status"] == 0)[:, None], ws["y"][:, :2], torch.zeros_like(ws["y"][:, :2])))
    torch.cuda.synchronize(dev)
    if total == warmup + steps:
        if barrier is not None:
            barrier()
        elapsed = time.perf_counter() - t0
    timed = range(warmup, warmup + steps)
    spans = sorted((ev_base.elapsed_time(a_), ev_base.elapsed_time(b_)) for t in timed for a_, b_ in evp[t])
    busy, ca, cb = 0.0, spans[0][0], spans[0][1]
    for a_, b_ in spans[1:]:
        if a_ > cb:
            busy += cb - ca
            ca, cb = a_, b_
        else:
            cb = max(cb, b_)
    busy += cb - ca
    rts = [evr[k][0].elapsed_time(evr[k][1]) for k in evr if warmup <= k[0] < warmup + steps]
    isz = p["X"].element_size()
    n_refits = sum(1 for (t, c) in refit_log if warmup <= t < warmup + steps)
    report = dict(schedule=schedule, data="loop", batch=Bt, parts=parts, stagger=bool(stagger), max_train=max_train, points_after_refit=window,
                  steps=steps, warmup=warmup, refit_every=refit_every, dt=dt, shift_invariant=bool(shift_invariant), dtype=str(dtype),
                  retry_levels=retry_levels, query_shift_invariant=bool(query_shift_invariant and shift_invariant), fit_iters=fit_iters,
                  factor_dtype=str(factor_dtype) if factor_dtype is not None else str(dtype), min_jitter_level=min_jitter_level, seconds=elapsed, ms_per_step=elapsed / steps * 1e3,
                  instance_steps_per_s=Bt * steps / elapsed, part_refits_in_timed_region=n_refits,
                  pass_busy_ms_per_step=busy / steps, refit_ms_per_part_refit=(sum(rts) / len(rts)) if rts else None,
                  refit_failures_after_retries=int(sum(int(pt.fail_count) for pt in P)) + (sum(g.rgp.count_drop_failures() for g in P) if online else 0),
                  instances_factored_per_retry_level=[int(v) for v in sum((pt.rgp.retry_counts if online else pt.retry_counts) for pt in P).tolist()],
                  jitter_level_max=float(max(float((pt.rgp.jitter_level if online else pt.level).max()) for pt in P)),
                  solver_optimal_fraction=float((ws["status"] == 0).float().mean()))
    if not online:
        pass_bytes = isz * (window * (window + 1) // 2 + 2 * window * n + window * (1 + m)) * Bt
        report["roofline"] = {"pass": dict(bound="hbm", kernel="posterior_step_kernel<%s, 3, 4, 0, 1, false, 0>" % ("float" if isz == 4 else "double"),
                                           algorithmic_bytes_per_step=pass_bytes, achieved=pass_bytes / (busy / steps * 1e-3) / 1e9, peak=8000.0,
                                           unit="GB/s", frac=pass_bytes / (busy / steps * 1e-3) / 1e9 / 8000.0, traffic=None,
                                           note="union of the part batches' pass intervals (HIP events on their streams); a part's refit on its own "
                                                "stream runs beside the other parts' passes and slows them -- this is the loop's rate, not the kernel's alone")}
    # ---- the final model, for the parity checks
    final = dict(hyper={k: v.clone() for k, v in hyper.items()}, xq_check=p["xq"], x=x, ws=ws)
    rowsX, rowsUH, rowsY, rowsJ, Mks, Bks = [], [], [], [], [], []
    for c, pt in enumerate(P):
        xqc = p["xq"][pt.sl].contiguous()
        with torch.cuda.stream(streams[c]):
            if online:
                g = pt.rgp
                rowsX.append(g.X[:, :g.N].clone()); rowsUH.append(g._rUH[:, :g.N].clone())
                rowsY.append(g._rY[:, :g.N].clone()); rowsJ.append(g._rJ[:, :g.N].clone())
                Mk, Bk = g.posterior(xqc)
            else:
                sl_ = slice(pt.lo, pt.lo + window)
                rowsX.append(pt.X[:, sl_].clone()); rowsUH.append(pt.UH[:, sl_].clone())
                rowsY.append(pt.Y[:, sl_].clone()); rowsJ.append(pt.J[:, sl_].clone())
                b_ = pt.bufs[pt.cur]
                Mk, Bk = ops.posterior_step(b_["Lop"], b_["Vw"], b_["X"], b_["UHB"], pt.hp["ell"], pt.hp["s2"], pt.hp["Bm"], pt.hp["M0"], xqc)
            Mks.append(Mk); Bks.append(Bk)
    torch.cuda.synchronize(dev)
    final["rows"] = [dict(X=rowsX[c], UH=rowsUH[c], Y=rowsY[c], jitter=rowsJ[c]) for c in range(parts)]   # (parts may hold different N)
    final["bounds"] = bounds
    final["posterior"] = (torch.cat(Mks, 0), torch.cat(Bks, 0))
    final["stream_rows"] = dict(X=Xall, UH=UHall, Y=Yall, jitter=Jall, window=window)
    if record_states:
        final["states"] = dict(x=torch.stack(xs_log, 1), u=torch.stack(us_log, 1))
    return report, final


def final_model_vs_fp64_refit(final, sample=64):
    """Self-check of `self_learning_closed_loop`'s final model: per part, a from-scratch fp64 refit ON THE DEVICE of the rows the
    model holds (with the jitter every point ended up with) against the model's own posterior at `xq_check`; max deviation over
    `sample` instances per part, relative to max(1, |M_k|) and the prior scale.  (The CPU-oracle check lives in tests/.)"""
    hyper, xq = final["hyper"], final["xq_check"]
    Mk_all, Bk_all = final["posterior"]
    worst = dict(Mk=0.0, Bk=0.0, refit_failures=0, instances=0)
    f64 = lambda t: t.double().contiguous()
    for (lo, hi), rows in zip(final["bounds"], final["rows"]):
        Bt = hi - lo
        idx = torch.linspace(0, Bt - 1, min(sample, Bt), device=xq.device).long()
        hp = {k: f64(hyper[k][lo:hi][idx]) for k in ("Bm", "ell", "s2", "M0")}
        X, UH, Y, J = (f64(rows[k][idx]) for k in ("X", "UH", "Y", "jitter"))
        Lr, UHBr, info, _ = ops.refit(X, UH, hp["Bm"], hp["ell"], hp["s2"], J)
        Vr, _ = ops.potrs(Lr, Y, UH, hp["M0"], want_alpha=False)
        Mr, Br = ops.posterior_step(Lr, Vr, X, UHBr, hp["ell"], hp["s2"], hp["Bm"], hp["M0"], f64(xq[lo:hi][idx]))
        prior = (hp["s2"][:, None, None] * hp["Bm"]).abs().amax(dim=(1, 2))
        ok = info == 0
        worst["refit_failures"] += int((~ok).sum())
        worst["instances"] += int(idx.numel())
        if bool(ok.any()):
            dM = ((f64(Mk_all[lo:hi][idx]) - Mr).abs().amax(dim=(1, 2)) / Mr.abs().amax(dim=(1, 2)).clamp(min=1.0))[ok].max()
            dB = ((f64(Bk_all[lo:hi][idx]) - Br).abs().amax(dim=(1, 2)) / prior)[ok].max()
            worst["Mk"], worst["Bk"] = max(worst["Mk"], float(dM)), max(worst["Bk"], float(dB))
    return worst
