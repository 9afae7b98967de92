"""Monte-Carlo safety rollouts (BASELINE config 4) and online-GP growth (config 5) drivers.

Batched counterpart of the reference's `unicycle_bayes_cbf_safe_obstacle` recipe
(unicycle_move_to_pose.py:1887-1928: fixed-kernel Ackermann model, CLFCartesian Kp=[.9,1.5,0],
two obstacles at mid path with weights [.7,.3], gamma 5, max_risk 0.01) run as Bt independent
closed loops from perturbed start states: `sample_generator_trajectory` (sampling.py:68-74) for
every trajectory at once, sharded over GPUs by `distributed.shard_range`, statistics reduced once
at the end."""
import math

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
                     reserved=True, fused=True, window=None):
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
        rgp = ops.ReservedGP(Lop, Vw, X, UHB, p["ell"], p["s2"], p["Bm"], p["M0"], N1) if reserved else None
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
                                   "one pass) + gp_append_rows" % ("double" if isz == 8 else "float", 1 + m),
                                   algorithmic_bytes_per_launch=byt / k, achieved=gbs, peak=8000.0, unit="GB/s",
                                   frac=None if gbs is None else gbs / 8000.0, traffic=None,
                                   how="sum over the segment's appends of Bt x online_pass_bytes(N) / sum of the HIP-event "
                                       "intervals around append(+query) [ms = append_ms]")
        segs.append(seg)
    out = dict(batch=Bt, N0=N0, N1=N1, dtype=str(dtype),
               storage=("reserved (in place)" + (", posterior query and append on one pass" if (fused and with_control) else ""))
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
