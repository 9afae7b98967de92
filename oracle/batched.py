"""Oracle, vectorised over instances (test infrastructure; also the "vectorised CPU baseline" of SURVEY 8d).

The same arithmetic as the scalar oracle modules, with a leading batch axis and torch CPU tensors (the reference's own
ops: `torch.linalg.cholesky`, triangular solves -- control_affine_model.py:366-377, 1051-1091) so that the host's BLAS
threads are used:

    posterior_step   gp_posterior.posterior_step          (control_affine_model.py:1051-1091, b = 1 per instance)
    constraint_rows  control_step.constraint_rows         (unicycle_move_to_pose.py:522-696, 880-906)
    reldeg1_cones    cbc.reldeg1_terms + convert_cbc_terms_to_socp_terms   (cbc2.py:7-23, :837-878)
    clf_cbf_socp     socp.clf_cbf_socp / socp.coneqp      (unicycle_move_to_pose.py:926-953; cvxopt coneqp)

Every function is checked against its scalar twin in tests/test_oracle_batched.py.  Never imported by the product.
"""
import math

import torch

F64 = torch.float64


# ------------------------------------------------------------------------------------------------ posterior
def refit(X, UH, Xdot, Bm, ell, s2, M0, jitter):
    """L = chol(k(X,X) o (UH B UH') + diag(jitter)), Vw = L^-1 (Xdot - UH M0), UHB = UH B   [B, ...]."""
    d = (X[:, :, None, :] - X[:, None, :, :]) / ell[:, None, None, :]
    Kb = s2[:, None, None] * torch.exp(-0.5 * (d * d).sum(-1)) * (UH @ Bm @ UH.transpose(1, 2))
    Kb = Kb + torch.diag_embed(jitter)
    L = torch.linalg.cholesky(Kb)
    Vw = torch.linalg.solve_triangular(L, Xdot - UH @ M0, upper=False)
    return L, Vw, UH @ Bm


def posterior_step(L, Vw, X, UHB, ell, s2, Bm, M0, xq):
    """(Mk[B,n,C], Bk[B,C,C]) at one query per instance: Phi = diag(k*) UHB, W = L^-1 Phi, Mk = M0' + Vw'W,
    Bk = s2 Bm - W'W."""
    z = (X - xq[:, None, :]) / ell[:, None, :]
    kstar = s2[:, None] * torch.exp(-0.5 * (z * z).sum(-1))
    W = torch.linalg.solve_triangular(L, kstar[:, :, None] * UHB, upper=False)
    Mk = M0.transpose(1, 2) + Vw.transpose(1, 2) @ W
    Bk = s2[:, None, None] * Bm - W.transpose(1, 2) @ W
    return Mk, Bk


# ------------------------------------------------------------------------------------------------ task rows
def _wrap(a):
    return torch.remainder(a + math.pi, 2 * math.pi) - math.pi


def constraint_rows(x, plan, dot_plan, Kp, clf_gamma, centers, radii, tw, gammas):
    """(grad[B,K,3], const[B,K], sign[K]) -- CLC row 0 (CLFCartesian), obstacle CBC rows 1.. (ObstacleCBF)."""
    xd, yd = plan[:, 0] - x[:, 0], plan[:, 1] - x[:, 1]
    rho2 = xd * xd + yd * yd
    phi = torch.atan2(yd, xd)
    alpha, beta = _wrap(x[:, 2] - phi), _wrap(plan[:, 2] - phi)
    sa, sb = torch.sin(alpha), torch.sin(beta)
    V = 0.5 * Kp[0] * rho2 + Kp[1] * (1 - torch.cos(alpha)) + Kp[2] * (1 - torch.cos(beta))
    gx = -Kp[0] * xd - Kp[1] * sa * yd / rho2 - Kp[2] * sb * yd / rho2
    gy = -Kp[0] * yd + Kp[1] * sa * xd / rho2 + Kp[2] * sb * xd / rho2
    gth = Kp[1] * sa
    ggx = Kp[0] * xd + Kp[1] * sa * yd / rho2 + Kp[2] * sb * yd / rho2
    ggy = Kp[0] * yd - Kp[1] * sa * xd / rho2 - Kp[2] * sb * xd / rho2
    ggth = Kp[2] * sb
    grads = [torch.stack([gx, gy, gth], -1)]
    consts = [ggx * dot_plan[:, 0] + ggy * dot_plan[:, 1] + ggth * dot_plan[:, 2] + clf_gamma * V]
    Kob = radii.shape[1]
    for k in range(Kob):
        g = x[:, :2] - centers[:, k]
        r2 = (g * g).sum(-1)
        rn = torch.sqrt(r2)
        th = x[:, 2]
        h = tw[0] * (r2 - radii[:, k] ** 2) + tw[1] * (torch.cos(th) * g[:, 0] + torch.sin(th) * g[:, 1]) / rn
        al = torch.atan2(g[:, 1], g[:, 0])
        s_ = torch.sin(al - th)
        grad = torch.stack([tw[0] * 2 * g[:, 0] + tw[1] * s_ * g[:, 1] / r2,
                            tw[0] * 2 * g[:, 1] - tw[1] * s_ * g[:, 0] / r2,
                            tw[1] * s_], -1)                 # -sin(th - al) = sin(al - th)
        grads.append(grad)
        consts.append(gammas[k] * h)
    sign = torch.tensor([-1.0] + [1.0] * Kob, dtype=x.dtype)
    return torch.stack(grads, 1), torch.stack(consts, 1), sign


def reldeg1_cones(Mk, Bk, A, grad, const, sign, ghat):
    """Rel-degree-1 terms and their cone form for every (instance, constraint): returns (cA[B,K,C,m], cb[B,K,C],
    cc[B,K,m], cd[B,K], ok[B,K]) with Asq = L L', cA = L'[:,1:], cb = L'[:,0]  (fhat = 0: the Ackermann prior)."""
    sg = sign[None, :, None]
    cc = sg * torch.einsum("bnm,bkn->bkm", ghat + Mk[:, :, 1:], grad)
    cd = sign[None, :] * (torch.einsum("bkn,bn->bk", grad, Mk[:, :, 0]) + const)
    a_h = torch.einsum("bkn,bnp,bkp->bk", grad, A, grad)
    Asq = a_h[:, :, None, None] * Bk[:, None, :, :]
    L, info = torch.linalg.cholesky_ex(Asq)
    Lt = L.transpose(-1, -2)
    return Lt[..., :, 1:], Lt[..., :, 0], cc, cd, info == 0


def ackermann_g(x, L_mean):
    z, o = torch.zeros_like(x[:, 2]), torch.ones_like(x[:, 2])
    return torch.stack([torch.stack([torch.cos(x[:, 2]), z], -1), torch.stack([torch.sin(x[:, 2]), z], -1),
                        torch.stack([z, o / L_mean], -1)], 1)


# ------------------------------------------------------------------------------------------------ cone program
def _jdot(u, v):
    return u[..., 0] * v[..., 0] - (u[..., 1:] * v[..., 1:]).sum(-1)


def _max_step(x):
    """[B]: -min 'eigenvalue' over the K cones of x[B,K,D]."""
    return (torch.linalg.vector_norm(x[..., 1:], dim=-1) - x[..., 0]).amax(dim=1)


def _sprod(x, y):
    out = torch.empty_like(x)
    out[..., 0] = (x * y).sum(-1)
    out[..., 1:] = x[..., :1] * y[..., 1:] + y[..., :1] * x[..., 1:]
    return out


def _sinv(lam, x):
    det = _jdot(lam, lam)
    lx = (lam[..., 1:] * x[..., 1:]).sum(-1)
    y0 = (lam[..., 0] * x[..., 0] - lx) / det
    out = torch.empty_like(x)
    out[..., 0] = y0
    out[..., 1:] = (x[..., 1:] - y0[..., None] * lam[..., 1:]) / lam[..., :1]
    return out


def _nt(s, z):
    """Per-cone Nesterov-Todd scaling: (W[B,K,D,D], Winv) with W z = Winv s."""
    D = s.shape[-1]
    sn, zn = torch.sqrt(_jdot(s, s)), torch.sqrt(_jdot(z, z))
    sb, zb = s / sn[..., None], z / zn[..., None]
    gamma = torch.sqrt((1.0 + (sb * zb).sum(-1)) / 2.0)
    w = torch.empty_like(s)
    w[..., 0] = (sb[..., 0] + zb[..., 0]) / (2 * gamma)
    w[..., 1:] = (sb[..., 1:] - zb[..., 1:]) / (2 * gamma[..., None])
    beta = torch.sqrt(sn / zn)
    eye = torch.eye(D - 1, dtype=s.dtype)
    out = []
    for sign, scale in ((1.0, beta), (-1.0, 1.0 / beta)):
        Wk = torch.empty(*s.shape, D, dtype=s.dtype)
        Wk[..., 0, 0] = w[..., 0]
        Wk[..., 0, 1:] = sign * w[..., 1:]
        Wk[..., 1:, 0] = sign * w[..., 1:]
        Wk[..., 1:, 1:] = eye + w[..., 1:, None] * w[..., None, 1:] / (1.0 + w[..., 0])[..., None, None]
        out.append(scale[..., None, None] * Wk)
    return out


def _scale2(lam, x):
    nrm = torch.sqrt(_jdot(lam, lam))
    lb = lam / nrm[..., None]
    lx = (lb[..., 1:] * x[..., 1:]).sum(-1)
    out = torch.empty_like(x)
    out[..., 0] = (lb[..., 0] * x[..., 0] - lx) / nrm
    out[..., 1:] = (x[..., 1:] + (-x[..., 0] + lx / (1.0 + lb[..., 0]))[..., None] * lb[..., 1:]) / nrm[..., None]
    return out


def coneqp_soc(Pd, q, G, h, maxiters=100, abstol=1e-9, reltol=1e-9, feastol=1e-9):
    """socp.coneqp for B programs at once, P = diag(Pd[B,nv]), K second-order cones of equal dimension D:
    G[B,K,D,nv], h[B,K,D].  Same algorithm, same constants, same stopping tests; instances that have stopped are
    frozen.  Returns dict(x[B,nv], status[B] (0 optimal, 1 'unknown': iteration limit or a failed factorisation, 2 diverged),
    iterations[B])."""
    B, K, D, nv = G.shape
    dt = G.dtype
    e = torch.zeros(K, D, dtype=dt)
    e[:, 0] = 1.0
    resx0 = torch.clamp(torch.linalg.vector_norm(q, dim=-1), min=1.0)
    resz0 = torch.clamp(torch.linalg.vector_norm(h.reshape(B, -1), dim=-1), min=1.0)
    Gf = G.reshape(B, K * D, nv)
    hf = h.reshape(B, K * D)
    # initial point: (P + G'G) x = G'h - q, z = G x - h, s = -z, shifted into the cone
    H0 = torch.diag_embed(Pd) + Gf.transpose(1, 2) @ Gf
    x = torch.linalg.solve(H0, (Gf.transpose(1, 2) @ hf[..., None])[..., 0] - q)
    z = ((Gf @ x[..., None])[..., 0] - hf).reshape(B, K, D)
    s = -z.clone()
    nrm_s = torch.clamp(torch.linalg.vector_norm(s.reshape(B, -1), dim=-1), min=1.0)
    ts, tz = _max_step(s), _max_step(z)
    s = s + torch.where(ts >= -1e-8 * nrm_s, 1.0 + ts, torch.zeros_like(ts))[:, None, None] * e
    z = z + torch.where(tz >= -1e-8 * nrm_s, 1.0 + tz, torch.zeros_like(tz))[:, None, None] * e
    M, Minv = _nt(s, z)
    lam = (M @ z[..., None])[..., 0]
    status = torch.ones(B, dtype=torch.int64)
    iters = torch.zeros(B, dtype=torch.int64)
    live = torch.ones(B, dtype=torch.bool)
    for it in range(maxiters + 1):
        Gt = Minv @ G                                                   # [B,K,D,nv]
        s = (M @ lam[..., None])[..., 0]
        z = (Minv.transpose(-1, -2) @ lam[..., None])[..., 0]
        f0 = (x * (0.5 * Pd * x + q)).sum(-1)
        rx = Pd * x + q + torch.einsum("bkdi,bkd->bi", G, z)
        rz = torch.einsum("bkdi,bi->bkd", G, x) + s - h
        rzt = (Minv @ rz[..., None])[..., 0]
        resx = torch.linalg.vector_norm(rx, dim=-1)
        resz = torch.linalg.vector_norm(rz.reshape(B, -1), dim=-1)
        gap = (lam * lam).sum((-1, -2))
        dcost = f0 + (lam * rzt).sum((-1, -2)) - gap
        relgap = torch.where(f0 < 0, gap / -f0, torch.where(dcost > 0, gap / dcost, torch.full_like(gap, math.inf)))
        pres, dres = resz / resz0, resx / resx0
        conv = (pres <= feastol) & (dres <= feastol) & ((gap <= abstol) | (relgap <= reltol))
        div = ~torch.isfinite(gap + resx + resz) | (x.abs().amax(-1) > 1e12)
        newly = live & (conv | div)
        status[live & conv] = 0
        status[live & div & ~conv] = 2
        iters[newly] = it
        live = live & ~newly
        if it == maxiters or not bool(live.any()):
            iters[live] = it
            break
        H = torch.diag_embed(Pd) + torch.einsum("bkdi,bkdj->bij", Gt, Gt)
        Hc, info = torch.linalg.cholesky_ex(H)
        failed = live & (info != 0)          # the scalar solver stops here with status 'unknown'
        iters[failed] = it
        live = live & ~failed
        Hc = torch.where((info != 0)[:, None, None], torch.eye(nv, dtype=dt).expand(B, nv, nv), Hc)
        lsq = _sprod(lam, lam)
        mu = gap / K
        sigma = torch.zeros(B, dtype=dt)
        corr = torch.zeros_like(lam)
        step = torch.ones(B, dtype=dt)
        for i in (0, 1):
            c = _sinv(lam, -lsq - corr + (sigma * mu)[:, None, None] * e)
            rhs = -rx + torch.einsum("bkdi,bkd->bi", Gt, -rzt - c)
            dx = torch.cholesky_solve(rhs[..., None], Hc)[..., 0]
            t = torch.einsum("bkdi,bi->bkd", Gt, dx) + rzt
            dzt, dst = t + c, -t
            if i == 0:
                corr = _sprod(dst, dzt)
                dsdz = (dst * dzt).sum((-1, -2))
            tm = torch.clamp(torch.maximum(_max_step(_scale2(lam, dst)), _max_step(_scale2(lam, dzt))), min=0.0)
            cap = 1.0 if i == 0 else 0.99
            step = torch.where(tm == 0, torch.ones_like(tm), torch.clamp(cap / tm, max=1.0))
            if i == 0:
                sigma = torch.clamp(1.0 - step + dsdz / gap * step ** 2, 0.0, 1.0) ** 3
        upd = live[:, None]
        x = torch.where(upd, x + step[:, None] * dx, x)
        st_, zt_ = lam + step[:, None, None] * dst, lam + step[:, None, None] * dzt
        What, Whatinv = _nt(st_, zt_)
        lam_new = (What @ zt_[..., None])[..., 0]
        u4 = live[:, None, None, None]
        lam = torch.where(live[:, None, None], lam_new, lam)
        M = torch.where(u4, M @ What, M)
        Minv = torch.where(u4, Whatinv @ Minv, Minv)
    return dict(x=x, status=status, iterations=iters)


def clf_cbf_socp(w, r, cA, cb, cc, cd, rho, relax_mask, maxiters=100):
    """socp.clf_cbf_socp for B instances: w[B,m+1], r[B,m], cones (cA[B,K,C,m], cb[B,K,C], cc[B,K,m], cd[B,K]),
    rho[B], relax_mask[K]."""
    B, K, C, m = cA.shape
    nv = m + 1
    Pd = 2.0 * w
    q = torch.zeros(B, nv, dtype=w.dtype)
    q[:, :m] = -2.0 * w[:, :m] * r
    G = torch.zeros(B, K, C + 1, nv, dtype=w.dtype)
    G[:, :, 0, :m] = -cc
    G[:, :, 0, m] = -relax_mask[None, :]
    G[:, :, 1:, :m] = -rho[:, None, None, None] * cA
    h = torch.cat([cd[..., None], rho[:, None, None] * cb], dim=-1)
    return coneqp_soc(Pd, q, G, h, maxiters=maxiters)


def control_step(L, Vw, X, UHB, ell, s2, Bm, M0, A, x, plan, dot_plan, Kp, clf_gamma, centers, radii, tw, gammas,
                 L_mean, w, r, rho, relax_mask, dt=0.0, L_true=1.0):
    """control_step.control_step for B instances with learned models (L, Vw, ... as `refit` returns them)."""
    Mk, Bk = posterior_step(L, Vw, X, UHB, ell, s2, Bm, M0, x)
    grad, const, sign = constraint_rows(x, plan, dot_plan, Kp, clf_gamma, centers, radii, tw, gammas)
    ghat = ackermann_g(x, L_mean)
    cA, cb, cc, cd, ok = reldeg1_cones(Mk, Bk, A, grad, const, sign, ghat)
    cA, cb = torch.nan_to_num(cA), torch.nan_to_num(cb)
    sol = clf_cbf_socp(w, r, cA, cb, cc, cd, rho, relax_mask)
    status = torch.where(ok.all(dim=1), sol["status"], torch.full_like(sol["status"], 3))
    u = sol["x"][:, :2]
    solved = (status == 0)[:, None]
    gt = ackermann_g(x, L_true)
    x_next = torch.where(solved & (dt > 0), x + torch.einsum("bnm,bm->bn", gt, u) * dt, x)
    return dict(y=sol["x"], status=status, iterations=sol["iterations"], Mk=Mk, Bk=Bk, x_next=x_next)
