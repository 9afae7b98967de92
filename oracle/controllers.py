"""Oracle: the generic (pendulum-style) controllers of bayes_cbf/controllers.py (test infrastructure).

numpy restatement of how `SOCPController` / `QPController` turn the quadratic terms of a constraint,
    mean(u) = bfe'u + e,        var(u) = u'V u + bfv'u + v        (cbc2_quadratic_terms, cbc2.py:7-23)
into rows of a cone program over y = [y_1; rho; u] (extravars = 2) or y = [rho; u] (extravars = 1).
Pinned by tests/golden/controllers_*.npz, recorded from the executed reference
(tests/golden/gen_golden.py `gen_controllers`), incl. instances whose Asq is indefinite (eigen fallback).
"""
import math

import numpy as np

from . import socp as _socp


def _asq(V, bfv, v):
    m = len(bfv)
    Asq = np.empty((m + 1, m + 1))
    Asq[0, 0] = v
    Asq[0, 1:] = np.asarray(bfv) / 2.0
    Asq[1:, 0] = np.asarray(bfv) / 2.0
    Asq[1:, 1:] = V
    return Asq


def _chol(Asq):
    """numpy raises LinAlgError where the reference's torch.cholesky raised '... singular U.'"""
    if not np.all(np.isfinite(Asq)):
        raise np.linalg.LinAlgError("non-finite")
    return np.linalg.cholesky(Asq)


def socp_objective(u0, ctrl_reg, relax_weight, extravars=2, yidx=0):
    """controllers.py:396-420  |R y + h| <= a'y + b with R = [[0, sqrt(lambda), 0], [0, 0, sqrt(Q) I]],
    h = [0; -sqrt(Q) u0], a = e_yidx, b = 0."""
    u0 = np.asarray(u0, dtype=np.float64)
    m = len(u0)
    assert extravars >= 2 and yidx < extravars
    sq = math.sqrt(ctrl_reg)
    R = np.zeros((m + 1, m + extravars))
    h = np.zeros(m + 1)
    R[0, 1] = math.sqrt(relax_weight)
    R[1:, extravars:] = sq * np.eye(m)
    h[1:] = -sq * u0
    a = np.zeros(m + extravars)
    a[yidx] = 1.0
    return R, h, a, 0.0


def convert_cbc_terms_to_socp_terms(bfe, e, V, bfv, v, extravars):
    """controllers.py:423-482 (static method; used by _socp_stability :484-500 and QPController._qp_stability
    :614-629): Asq = L L' (retried once with + 1e-3 I, :464-469); A[:, ev:] = L'[:, 1:], b = L'[:, 0],
    c = [.., 1 at ev-1, bfe], d = e."""
    m = len(bfe)
    Asq = _asq(V, bfv, v)
    try:
        L = _chol(Asq)
    except np.linalg.LinAlgError:
        L = _chol(Asq + 1e-3 * np.eye(m + 1))
    A = np.zeros((m + 1, m + extravars))
    A[:, extravars:] = L.T[:, 1:]
    bfb = L.T[:, 0].copy()
    bfc = np.zeros(m + extravars)
    assert extravars >= 1
    bfc[extravars - 1] = 1.0
    bfc[extravars:] = bfe
    return A, bfb, bfc, float(e)


def socp_safety(bfe, e, V, bfv, v, factor, extravars):
    """controllers.py:502-540: A[:, ev:] = L[:, 1:], b = L[:, 0] with the LOWER Cholesky factor L of Asq (the
    reference reads L here, not L' as in convert_cbc_terms_to_socp_terms); when the factorisation fails,
    L = sqrt(max(Lambda, 0)) V' from symeig (:528-530, ascending eigenvalues); returns (factor A, factor b, c, e)
    with c = [0.., bfe]."""
    m = len(bfe)
    Asq = _asq(V, bfv, v)
    try:
        L = _chol(Asq)
    except np.linalg.LinAlgError:
        w, Q = np.linalg.eigh(Asq)
        L = np.sqrt(np.maximum(np.diag(w), 0.0)) @ Q.T
    A = np.zeros((m + 1, m + extravars))
    A[:, extravars:] = L[:, 1:]
    b = L[:, 0].copy()
    c = np.zeros(m + extravars)
    c[extravars:] = bfe
    return factor * A, factor * b, c, float(e)


def named_socp_constraints(u_ref, ctrl_reg, relax_weight, safety_terms, safety_factors, stability_terms=None,
                           extravars=2):
    """controllers.py:542-567: [Objective] + [Safety_i gt 0] + [Stability gt 0]; *_terms = (bfe, e, V, bfv, v)."""
    cons = [("Objective", socp_objective(u_ref, ctrl_reg, relax_weight, extravars=extravars))]
    for i, (terms, f) in enumerate(zip(safety_terms, safety_factors)):
        cons.append(("Safety_%d gt 0" % i, socp_safety(*terms, f, extravars)))
    if stability_terms is not None:
        cons.append(("Stability gt 0", convert_cbc_terms_to_socp_terms(*stability_terms, extravars)))
    return cons


def socp_controller_control(u_ref, ctrl_reg, relax_weight, safety_terms, safety_factors, stability_terms=None):
    """SOCPController.control (:569-591): min y_1 over y = [y_1; rho; u] subject to the named cones; returns
    (u, y, solver dict).  The reference solves with cvxpy/GUROBI; any accurate solver is a valid checker
    (strictly convex in (rho, u) through the epigraph cone)."""
    m = len(u_ref)
    cons = named_socp_constraints(u_ref, ctrl_reg, relax_weight, safety_terms, safety_factors, stability_terms)
    c = np.concatenate([[1.0, 0.0], np.zeros(m)])
    sol = _socp.optimizer_socp(c, cons)
    return sol["x"][2:], sol["x"], sol


def qp_controller_control(u_ref, ctrl_reg, relax_weight, stability_terms):
    """QPController.control (:631-662): min |A y|^2, A = diag(sqrt(lambda), sqrt(Q)..), (b = 0: u_ref does not enter
    the objective in the reference) s.t. 0 <= c'y + d with (c, d) from _qp_stability (:614-629), y = [rho; u]."""
    m = len(u_ref)
    A = np.zeros((1 + m, 1 + m))
    A[0, 0] = math.sqrt(relax_weight)
    A[1:, 1:] = math.sqrt(ctrl_reg) * np.eye(m)
    _, _, bfc, d = convert_cbc_terms_to_socp_terms(*stability_terms, 1)
    sol = _socp.optimizer_qp((A, np.zeros(1 + m)), [("Safety", (bfc, d))])
    return sol["x"][1:], sol["x"], sol
