"""Seeded synthetic workloads of the shapes BASELINE.json names.

Hyper-parameters are fixed, not fitted: ell_d = s2 = softplus(0) (gpytorch's initial values),
A = W_A W_A' + diag(softplus(v_A)), B likewise (rank-1 W as in the ...RankOne regressors the
demos use), M0 = 0.  The jitter eps = 1e-5 * U[0,1)^N is drawn here (the library never draws
randomness).  Generation is plain torch: it is input plumbing, not part of the timed path.
"""
import math

import torch

SOFTPLUS0 = math.log(2.0)


def _index_kernel(gen, Bt, k, rank, dtype, device):
    W = torch.randn(Bt, k, rank, generator=gen, dtype=torch.float64, device=device)
    v = torch.randn(Bt, k, generator=gen, dtype=torch.float64, device=device)
    M = W @ W.transpose(1, 2) + torch.diag_embed(torch.nn.functional.softplus(v))
    return M.to(dtype)


def make_instances(Bt, N, n, m, dtype=torch.float32, device="cuda", seed=1234, variant="dense"):
    """Independent GP instances (regime I).  variant 'dense': x,y ~ U[-3,0], theta ~ U[-pi,pi]
    (well conditioned); 'theta': shift-invariant training inputs [0,..,0,theta] as the reference
    trains the unicycle (unicycle_move_to_pose.py:326-330) -- numerically rank deficient."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    f64 = torch.float64
    X = torch.empty(Bt, N, n, dtype=f64, device=device)
    if variant == "theta":
        X.zero_()
    else:
        X.uniform_(-3.0, 0.0, generator=gen)
    X[..., n - 1].uniform_(-math.pi, math.pi, generator=gen)
    U = torch.randn(Bt, N, m, generator=gen, dtype=f64, device=device)
    scale = torch.tensor([2.0, math.pi, 1.0][:m], dtype=f64, device=device)
    U = U * scale
    UH = torch.cat([torch.ones(Bt, N, 1, dtype=f64, device=device), U], dim=2)
    # residual targets: smooth control-affine function of the state + noise
    Wf = torch.randn(Bt, n, n, generator=gen, dtype=f64, device=device) * 0.3
    Wg = torch.randn(Bt, n, n, m, generator=gen, dtype=f64, device=device) * 0.1
    Xdot = (torch.sin(X @ Wf.transpose(1, 2))
            + torch.einsum("bin,bknm,bim->bik", torch.cos(X), Wg, U)
            + 1e-3 * torch.randn(Bt, N, n, generator=gen, dtype=f64, device=device))
    A = _index_kernel(gen, Bt, n, n, f64, device)
    Bm = _index_kernel(gen, Bt, 1 + m, 1, f64, device)
    ell = torch.full((Bt, n), SOFTPLUS0, dtype=f64, device=device)
    s2 = torch.full((Bt,), SOFTPLUS0, dtype=f64, device=device)
    M0 = torch.zeros(Bt, 1 + m, n, dtype=f64, device=device)
    jitter = 1e-5 * torch.rand(Bt, N, generator=gen, dtype=f64, device=device)
    jitter2 = 1e-5 * torch.rand(Bt, 1 + m, generator=gen, dtype=f64, device=device)
    # one query per instance inside the training bounding box
    lo, hi = X.amin(dim=1), X.amax(dim=1)
    xq = lo + (hi - lo) * torch.rand(Bt, n, generator=gen, dtype=f64, device=device)
    out = dict(X=X, U=U, UH=UH, Xdot=Xdot, A=A, Bm=Bm, ell=ell, s2=s2, M0=M0, jitter=jitter, jitter2=jitter2, xq=xq)
    return {k: v.to(dtype).contiguous() for k, v in out.items()}


def make_unicycle_task(Bt, dtype=torch.float32, device="cuda", seed=99):
    """Per-instance start/goal, obstacles and planner targets for the unicycle constraints
    (CLFCartesian Kp=[.9,1.5,0], gamma 10; two ObstacleCBFs at mid path, weights [.7,.3], gamma 5;
    max_risk 0.01; cost weights .33 -- the saved-run recipe, unicycle_move_to_pose.py:1887-1928)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    f64 = torch.float64
    x0 = torch.tensor([-3.0, -1.0, -math.pi / 4], dtype=f64, device=device).expand(Bt, 3).clone()
    x0 += 0.1 * torch.randn(Bt, 3, generator=gen, dtype=f64, device=device)
    xg = torch.tensor([0.0, 0.0, math.pi / 4], dtype=f64, device=device).expand(Bt, 3).clone()
    R90 = torch.tensor([[0.0, -1.0], [1.0, 0.0]], dtype=f64, device=device)
    mid = (x0[:, :2] + xg[:, :2]) / 2
    off = (x0[:, :2] - xg[:, :2]) @ R90.T / 3
    centers = torch.stack([mid + off, mid - off], dim=1)
    radii = ((x0[:, :2] - xg[:, :2]).norm(dim=1) / 4).unsqueeze(1).expand(Bt, 2).clone()
    # current state a little way along the path; plan = look-ahead target towards the goal
    frac = torch.rand(Bt, 1, generator=gen, dtype=f64, device=device) * 0.3
    x = x0 + frac * (xg - x0) + 0.05 * torch.randn(Bt, 3, generator=gen, dtype=f64, device=device)
    plan = x0 + (frac + 0.1) * (xg - x0)
    dot_plan = (xg - x0) / 2.0
    out = dict(x0=x0, xg=xg, x=x, plan=plan, dot_plan=dot_plan, centers=centers, radii=radii,
               Kp=torch.tensor([0.9, 1.5, 0.0], dtype=f64, device=device),
               tw=torch.tensor([0.7, 0.3], dtype=f64, device=device),
               gammas=torch.tensor([5.0, 5.0], dtype=f64, device=device),
               sign=torch.tensor([-1.0, 1.0, 1.0], dtype=f64, device=device),
               relax_mask=torch.tensor([1.0, 0.0, 0.0], dtype=f64, device=device),
               w=torch.full((Bt, 3), 0.33, dtype=f64, device=device),
               r=torch.zeros(Bt, 2, dtype=f64, device=device),
               rho=torch.full((Bt,), 2.3263478740408408, dtype=f64, device=device))
    return {k: v.to(dtype).contiguous() for k, v in out.items()}
