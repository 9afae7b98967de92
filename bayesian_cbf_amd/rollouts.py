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
                                record=False, max_iters=30):
    """Run Bt closed loops for numSteps steps.  `gp`: dict from BatchedControlAffineGP.as_dict() (learned
    residual, one GP per trajectory) or None (fixed-kernel model M_k = 0, B_k = I, A = diag(kernel_diag_A)).
    Returns dict(stats..., x_final[Bt,3], traj (if record)).  Collectives: one, at the end."""
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
        Mk0 = torch.zeros(Bt, 3, 3, **f)
        Bk0 = torch.eye(3, **f).expand(Bt, 3, 3).contiguous()
    min_h = torch.full((Bt,), float("inf"), **f)
    cost = torch.zeros(Bt, **f)
    fails = torch.zeros(Bt, dtype=torch.int32, device=dev)
    gam = task["gammas"]
    traj = torch.empty(numSteps + 1, Bt, 3, **f) if record else None
    if record:
        traj[0] = x
    for t in range(numSteps):
        task["plan"] = planner.plan(t).to(**f).expand(Bt, 3).contiguous()
        task["dot_plan"] = planner.dot_plan(t).to(**f).expand(Bt, 3).contiguous()
        if gp is not None:
            ops.unicycle_control_step(gp, task, ws, x, dt=dt, L_true=L_true, L_mean=L_mean, max_iters=max_iters)
        else:
            ops.unicycle_constraints(x, task["plan"], task["dot_plan"], task["Kp"], 10.0, task["centers"], task["radii"],
                                     task["tw"], task["gammas"], L_mean, out=(ws["grad"], ws["cst"], ws["fhat"], ws["ghat"]))
            y, status, iters, _, _, _ = ops.cbc_socp(Mk0, Bk0, A, ws["grad"], ws["cst"], task["sign"], ws["fhat"],
                                                     ws["ghat"], task["w"], task["r"], task["relax_mask"], task["rho"],
                                                     max_iters=max_iters)
            ws["y"].copy_(y)
            ws["status"].copy_(status)
            bad = status != 0
            if bool(bad.any()):      # infeasible instances are masked (reference control), not fatal
                ws["y"][bad] = 0
            ops.unicycle_step(x, ws["y"][:, :2].contiguous(), dt, L_true)
        # safety bookkeeping: h_k(x_t) = cst_k / gamma_k for the obstacle rows (before the step)
        h = ws["cst"][:, 1:] / gam
        min_h = torch.minimum(min_h, h.min(dim=1).values)
        cost += (task["w"] * ws["y"] ** 2).sum(dim=1)
        fails += (ws["status"] != 0).to(torch.int32)
        if record:
            traj[t + 1] = x
    collided = (min_h < 0)
    stats = reduce_rollout_stats(collided.sum(), min_h.min(), cost.sum() / numSteps, (fails > 0).sum(), Bt)
    dist_to_goal = (x[:, :2] - xg[:2]).norm(dim=1)
    return dict(stats=stats, x_final=x, min_h=min_h, dist_to_goal=dist_to_goal, traj=traj)
