"""Oracle: the per-step conic program (test infrastructure).

The reference hands its program to third-party solvers that are not in its tree:
cvxpy 1.0.25 + gurobipy 9.1.2 (bayes_cbf/unicycle_move_to_pose.py:926-953, requirements.txt:6-7)
and cvxopt 1.2.3 `solvers.socp` (bayes_cbf/optimizers.py:66-73, pip-freeze.txt:5).
Every program on this path is a strictly convex QP / SOCP (or has a unique optimum), so any
accurate solver is a valid oracle.  This file restates the *published* algorithm of cvxopt's
`coneqp` (L. Vandenberghe, "The CVXOPT linear and quadratic cone program solvers", 2010:
primal-dual path following, Nesterov-Todd scaling, Mehrotra correction, step 0.99, sigma =
(1-step)^3) in dense numpy.  Pinned by the reference's own known answer
(tests/test_optimizers.py:6-26,28-119), by KKT residual checks and by the GUROBI outputs
logged in docs/saved-runs (tests/golden/saved_run_*.npz).
"""
import numpy as np

MAXITERS = 100
ABSTOL = 1e-9
RELTOL = 1e-9
FEASTOL = 1e-9
STEP = 0.99
EXPON = 3


# ------------------------------------------------------------------ cone algebra (dims: l, q[])
def _blocks(dims):
    off = dims.get('l', 0)
    out = []
    for k in dims.get('q', []):
        out.append((off, off + k))
        off += k
    return out


def _cone_dim(dims):
    return dims.get('l', 0) + sum(dims.get('q', []))


def _degree(dims):
    return dims.get('l', 0) + len(dims.get('q', []))


def _identity(dims):
    e = np.zeros(_cone_dim(dims))
    e[:dims.get('l', 0)] = 1.0
    for a, _ in _blocks(dims):
        e[a] = 1.0
    return e


def _jdot(u, v):
    return u[0] * v[0] - u[1:] @ v[1:]


def _max_step(x, dims):
    """-min 'eigenvalue' of x: x + t*e is in the cone iff t >= this."""
    l = dims.get('l', 0)
    t = [-x[:l].min()] if l else []
    for a, b in _blocks(dims):
        t.append(np.linalg.norm(x[a + 1:b]) - x[a])
    return max(t)


def _sprod(x, y, dims):
    """Jordan product x o y."""
    l = dims.get('l', 0)
    out = np.empty_like(x)
    out[:l] = x[:l] * y[:l]
    for a, b in _blocks(dims):
        out[a] = x[a:b] @ y[a:b]
        out[a + 1:b] = x[a] * y[a + 1:b] + y[a] * x[a + 1:b]
    return out


def _sinv(lmbda, x, dims):
    """Solve lmbda o y = x for y."""
    l = dims.get('l', 0)
    out = np.empty_like(x)
    out[:l] = x[:l] / lmbda[:l]
    for a, b in _blocks(dims):
        lam = lmbda[a:b]
        det = _jdot(lam, lam)
        lx = lam[1:] @ x[a + 1:b]
        y0 = (lam[0] * x[a] - lx) / det
        out[a] = y0
        out[a + 1:b] = (x[a + 1:b] - y0 * lam[1:]) / lam[0]
    return out


def _nt_scaling(s, z, dims):
    """Dense symmetric Nesterov-Todd scaling W and its inverse: W z = W^-1 s = lambda."""
    K = _cone_dim(dims)
    W = np.zeros((K, K))
    Winv = np.zeros((K, K))
    l = dims.get('l', 0)
    for i in range(l):
        W[i, i] = np.sqrt(s[i] / z[i])
        Winv[i, i] = 1.0 / W[i, i]
    for a, b in _blocks(dims):
        sk, zk = s[a:b], z[a:b]
        sn, zn = np.sqrt(_jdot(sk, sk)), np.sqrt(_jdot(zk, zk))
        sb, zb = sk / sn, zk / zn
        gamma = np.sqrt((1.0 + sb @ zb) / 2.0)
        w = np.empty(b - a)
        w[0] = (sb[0] + zb[0]) / (2 * gamma)
        w[1:] = (sb[1:] - zb[1:]) / (2 * gamma)
        beta = np.sqrt(sn / zn)
        for mat, sign, scale in ((W, 1.0, beta), (Winv, -1.0, 1.0 / beta)):
            Wk = np.empty((b - a, b - a))
            Wk[0, 0] = w[0]
            Wk[0, 1:] = sign * w[1:]
            Wk[1:, 0] = sign * w[1:]
            Wk[1:, 1:] = np.eye(b - a - 1) + np.outer(w[1:], w[1:]) / (1.0 + w[0])
            mat[a:b, a:b] = scale * Wk
    return W, Winv


def _scale2(lmbda, x, dims):
    """x := P(lmbda^-1/2) x so that lmbda + t*x_in in K  <=>  e + t*x_out in K."""
    l = dims.get('l', 0)
    out = np.empty_like(x)
    out[:l] = x[:l] / lmbda[:l]
    for a, b in _blocks(dims):
        lam = lmbda[a:b]
        nrm = np.sqrt(_jdot(lam, lam))
        lb = lam / nrm
        x0 = x[a]
        x1 = x[a + 1:b]
        lx = lb[1:] @ x1
        out[a] = (lb[0] * x0 - lx) / nrm
        out[a + 1:b] = (x1 + (-x0 + lx / (1.0 + lb[0])) * lb[1:]) / nrm
    return out


# ------------------------------------------------------------------ the solver
def coneqp(P, q, G, h, dims, maxiters=MAXITERS, abstol=ABSTOL, reltol=RELTOL, feastol=FEASTOL):
    """min 1/2 x'Px + q'x  s.t.  G x + s = h,  s in K = R_+^l x Q^{q_1} x ...

    The iterates are kept in scaled coordinates (st = M^-1 s, zt = M' z, both near the
    central path) and the scaling M is the accumulated product of per-iteration NT scalings
    -- the device solver uses the same bookkeeping, which avoids the s0^2-|s1|^2 cancellation
    of recomputing a scaling from unscaled boundary points.
    Returns dict(x, s, z, status, iterations, gap, pres, dres); status 'optimal', 'unknown'
    (iteration limit) or 'diverged' (infeasible program: the reference raises there,
    bayes_cbf/optimizers.py:74-86, unicycle_move_to_pose.py:954-964).
    """
    P = np.asarray(P, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    G = np.asarray(G, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    nv = q.shape[0]
    K = _cone_dim(dims)
    e = _identity(dims)
    resx0 = max(1.0, np.linalg.norm(q))
    resz0 = max(1.0, np.linalg.norm(h))

    # initial point: [P G'; G -I][x; z] = [-q; h], s = -z, then shift into the cone
    KKT = np.block([[P, G.T], [G, -np.eye(K)]])
    sol = np.linalg.solve(KKT, np.concatenate([-q, h]))
    x, z = sol[:nv], sol[nv:]
    s = -z.copy()
    ts = _max_step(s, dims)
    if ts >= -1e-8 * max(np.linalg.norm(s), 1.0):
        s = s + (1.0 + ts) * e
    tz = _max_step(z, dims)
    if tz >= -1e-8 * max(np.linalg.norm(z), 1.0):
        z = z + (1.0 + tz) * e

    M, Minv = _nt_scaling(s, z, dims)
    lmbda = M @ z                      # = Minv @ s
    status = 'unknown'
    gap = pres = dres = np.inf
    it = 0
    for it in range(maxiters + 1):
        Gt = Minv @ G
        f0 = 0.5 * x @ P @ x + q @ x
        s = M @ lmbda                              # unscaled iterates: used for the residuals only
        z = Minv.T @ lmbda
        rx = P @ x + q + G.T @ z
        rz = G @ x + s - h
        rzt = Minv @ rz
        resx, resz = np.linalg.norm(rx), np.linalg.norm(rz)
        gap = lmbda @ lmbda
        pcost = f0
        dcost = f0 + lmbda @ rzt - gap
        if pcost < 0.0:
            relgap = gap / -pcost
        elif dcost > 0.0:
            relgap = gap / dcost
        else:
            relgap = None
        pres, dres = resz / resz0, resx / resx0
        if pres <= feastol and dres <= feastol and (gap <= abstol or (relgap is not None and relgap <= reltol)):
            status = 'optimal'
            break
        if not np.isfinite(gap + resx + resz) or np.abs(x).max() > 1e12:
            status = 'diverged'     # primal or dual infeasible: the reference raises here
            break
        if it == maxiters:
            break
        H = P + Gt.T @ Gt
        try:
            Hc = np.linalg.cholesky(H)
        except np.linalg.LinAlgError:
            break
        lmbdasq = _sprod(lmbda, lmbda, dims)
        mu = gap / _degree(dims)
        sigma = 0.0
        dsa_dza = np.zeros(K)
        step = 1.0
        for i in (0, 1):
            c = _sinv(lmbda, -lmbdasq - dsa_dza + sigma * mu * e, dims)
            # [P G'; G -MM'][dx; dz] = [-rx; -rz - M c]
            dx = np.linalg.solve(Hc.T, np.linalg.solve(Hc, -rx + Gt.T @ (-rzt - c)))
            t = Gt @ dx + rzt
            dzt = t + c
            dst = -t
            if i == 0:
                dsa_dza = _sprod(dst, dzt, dims)
                dsdz = dst @ dzt
            ts = _max_step(_scale2(lmbda, dst, dims), dims)
            tz = _max_step(_scale2(lmbda, dzt, dims), dims)
            tm = max(0.0, ts, tz)
            if tm == 0.0:
                step = 1.0
            else:
                step = min(1.0, 1.0 / tm) if i == 0 else min(1.0, STEP / tm)
            if i == 0:
                sigma = min(1.0, max(0.0, 1.0 - step + dsdz / gap * step ** 2)) ** EXPON
        x = x + step * dx
        st = lmbda + step * dst
        zt = lmbda + step * dzt
        What, Whatinv = _nt_scaling(st, zt, dims)
        lmbda = What @ zt
        M = M @ What
        Minv = Whatinv @ Minv
    return dict(x=x, s=M @ lmbda, z=Minv.T @ lmbda, status=status, iterations=it, gap=gap,
                pres=pres, dres=dres)


# ------------------------------------------------------------------ reference adapters
def convert_socp_to_cvxopt_format(c, socp_constraints):
    """bayes_cbf/optimizers.py:6-39: |A u + b| <= c'u + d  ->  Gq = [-c'; -A], hq = [d; b]."""
    m = np.asarray(c).shape[-1]
    Gqs, hqs = [], []
    for _name, (A, bfb, bfc, d) in socp_constraints:
        A = np.asarray(A, dtype=np.float64)
        Gq = np.zeros((A.shape[0] + 1, m))
        Gq[0, :] = -np.asarray(bfc)
        Gq[1:, :] = -A
        hq = np.zeros((A.shape[0] + 1, 1))
        hq[0, 0] = np.asarray(d).reshape(())
        hq[1:, 0] = bfb
        Gqs.append(Gq)
        hqs.append(hq)
    return c, Gqs, hqs


def optimizer_socp(linear_objective, socp_constraints, regularization=0.0):
    """min c'y  s.t. |A_k y + b_k| <= c_k'y + d_k   (bayes_cbf/optimizers.py:42-89, 91-102)."""
    c, Gqs, hqs = convert_socp_to_cvxopt_format(np.asarray(linear_objective, dtype=np.float64),
                                                socp_constraints)
    G = np.vstack(Gqs)
    h = np.concatenate([hq[:, 0] for hq in hqs])
    nv = G.shape[1]
    P = regularization * np.eye(nv)
    # cvxopt's conelp handles P = 0 through a self-dual embedding; with these well-posed test
    # programs G has full column rank so H = G' W^-2 G is positive definite and coneqp applies.
    return coneqp(P, c, G, h, dict(l=0, q=[g.shape[0] for g in Gqs]))


def optimizer_qp(quadratic_objective, linear_constraints):
    """min |A y + b|^2  s.t. 0 <= c_k'y + d_k   (bayes_cbf/optimizers.py:105-116)."""
    A, bfb = quadratic_objective
    A = np.asarray(A, dtype=np.float64)
    P = 2 * A.T @ A
    q = 2 * A.T @ np.asarray(bfb, dtype=np.float64)
    G = np.stack([-np.asarray(c, dtype=np.float64) for _n, (c, _d) in linear_constraints])
    h = np.array([float(np.asarray(d).reshape(())) for _n, (_c, d) in linear_constraints])
    return coneqp(P, q, G, h, dict(l=len(h), q=[]))


def clf_cbf_socp(weights, u_ref, cones, rho, relax_mask, maxiters=MAXITERS):
    """The program of ControllerCLFBayesian.control  (bayes_cbf/unicycle_move_to_pose.py:926-953).

    min sum_i w_i (u_i - r_i)^2 + w_relax relax^2
    s.t. c_k'u + d_k + relax_mask_k * relax >= rho * |A_k u + b_k|    for every cone k
    weights[m+1], u_ref[m], cones = [(A_k[(m+1),m], b_k[m+1], c_k[m], d_k)], relax_mask[K].
    Returns the coneqp dict with x = [u, relax].
    """
    w = np.asarray(weights, dtype=np.float64)
    m = len(u_ref)
    nv = m + 1
    P = 2.0 * np.diag(w)
    q = np.zeros(nv)
    q[:m] = -2.0 * w[:m] * np.asarray(u_ref, dtype=np.float64)
    Gs, hs, dimsq = [], [], []
    for (A, b, c, d), rm in zip(cones, relax_mask):
        A = np.asarray(A, dtype=np.float64)
        Gk = np.zeros((A.shape[0] + 1, nv))
        Gk[0, :m] = -np.asarray(c)
        Gk[0, m] = -float(rm)
        Gk[1:, :m] = -rho * A
        hk = np.concatenate([[float(np.asarray(d).reshape(()))], rho * np.asarray(b, dtype=np.float64)])
        Gs.append(Gk)
        hs.append(hk)
        dimsq.append(A.shape[0] + 1)
    return coneqp(P, q, np.vstack(Gs), np.concatenate(hs), dict(l=0, q=dimsq), maxiters=maxiters)


def kkt_residuals(P, q, G, h, dims, sol):
    """Stationarity / feasibility / complementarity residuals of a returned solution."""
    x, s, z = sol['x'], sol['s'], sol['z']
    return dict(stationarity=np.linalg.norm(P @ x + q + G.T @ z),
                primal=np.linalg.norm(G @ x + s - h),
                complementarity=abs(s @ z),
                s_cone=max(0.0, _max_step(s, dims)), z_cone=max(0.0, _max_step(z, dims)))
