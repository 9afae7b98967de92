"""Mirror of bayes_cbf/optimizers.py on the native batched cone solver (no cvxopt / cvxpy / GUROBI).

Same call signatures and return values; infeasible programs raise `InfeasibleProblemError` like
the reference (optimizers.py:3, 74-86)."""
import numpy as np
import torch

from . import ops


class InfeasibleProblemError(ValueError):
    pass


def convert_socp_to_cvxopt_format(c, socp_constraints):
    """|A u + b| <= c'u + d  ->  Gq = [-c'; -A], hq = [d; b]   (optimizers.py:6-39)."""
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


def _solve(P, q, G, h, l, qdims, device):
    dt = torch.float64
    dev = torch.device(device)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)[None].contiguous()
    x, status, iters = ops.coneqp(t(P), t(q), t(G), t(h), l, list(qdims))
    if int(status[0]) != 0:
        raise InfeasibleProblemError("Infeasible problem: solver status %d" % int(status[0]))
    return x[0].cpu().numpy()


def optimizer_socp_cvxopt(u0, linear_objective, socp_constraints, device="cuda"):
    """min c'u s.t. second-order cones (optimizers.py:42-89)."""
    c, Gqs, hqs = convert_socp_to_cvxopt_format(np.asarray(linear_objective, dtype=np.float64), socp_constraints)
    G = np.vstack(Gqs)
    h = np.concatenate([hq[:, 0] for hq in hqs])
    nv = G.shape[1]
    y = _solve(np.zeros((nv, nv)), c, G, h, 0, [g.shape[0] for g in Gqs], device)
    return y.astype(np.asarray(u0).dtype).reshape(-1)


def optimizer_socp_cvxpy(u0, linear_objective, socp_constraints, solver=None, device="cuda"):
    """Same program through the reference's cvxpy entry point (optimizers.py:91-102)."""
    return optimizer_socp_cvxopt(u0, linear_objective, socp_constraints, device=device)


def optimizer_qp_cvxpy(u0, quadratic_objective, linear_constraints, solver=None, device="cuda"):
    """min |A y + b|^2 s.t. 0 <= c'y + d   (optimizers.py:105-116)."""
    A, bfb = quadratic_objective
    A = np.asarray(A, dtype=np.float64)
    P = 2 * A.T @ A
    q = 2 * A.T @ np.asarray(bfb, dtype=np.float64)
    G = np.stack([-np.asarray(c, dtype=np.float64) for _n, (c, _d) in linear_constraints])
    h = np.array([float(np.asarray(d).reshape(())) for _n, (_c, d) in linear_constraints])
    return _solve(P, q, G, h, len(h), [], device).astype(np.asarray(u0).dtype)
