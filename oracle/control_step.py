"""Oracle: ONE control step of the unicycle Bayesian CLF-CBF controller, end to end (test infrastructure).

Restates `ControllerCLFBayesian.control` (bayes_cbf/unicycle_move_to_pose.py:926-995) for one state on top of the
other oracle modules, in the order the reference runs it:

    plan(t), dot_plan(t)                         planner.py:54-64 (inputs here)
    _clc_terms(x, plan)   -> cone (relaxed)      :880-899   (CLFCartesian :522-615, sign -1, + grad_goal V' xdot_plan)
    _cbcs(x)              -> one cone / obstacle :901-920   (ObstacleCBF :618-696)
       each through the learned model's posterior at x (control_affine_model.py:1051-1091, here `Mk`, `Bk`),
       cbc2_quadratic_terms (cbc2.py:7-23) and convert_cbc_terms_to_socp_terms (:837-878)
    min sum w_i (u_i - r_i)^2 + w_r relax^2  s.t. cones    :926-953   (GUROBI there, coneqp here)
    x+ = x + g(x; L_true) u dt                            :277-282   (sampling.py:68-74)

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline only -- never by the product path.
"""
import numpy as np

from . import cbc as ocbc
from . import socp as osocp
from . import unicycle as ouni


def constraint_rows(x, plan, dot_plan, Kp, clf_gamma, centers, radii, tw, gammas):
    """(grad[K,3], const[K], sign[K]) of the CLC (row 0) and the obstacle CBCs (rows 1..)."""
    clf = ouni.CLFCartesian(Kp)
    grad = [clf.grad_clf(x, plan)]
    const = [clf.grad_clf_wrt_goal(x, plan) @ np.asarray(dot_plan) + clf_gamma * clf.clf(x, plan)]     # :880-888
    sign = [-1.0]
    for k in range(len(radii)):
        ob = ouni.ObstacleCBF(centers[k], radii[k], tuple(tw))
        grad.append(ob.grad_cbf(x))
        const.append(gammas[k] * ob.cbf(x))                                                           # :901-906
        sign.append(1.0)
    return np.array(grad), np.array(const), np.array(sign)


def control_step(x, plan, dot_plan, Mk, Bk, A, Kp, clf_gamma, centers, radii, tw, gammas, L_mean, w, r, rho,
                 relax_mask=None, dt=0.0, L_true=1.0, maxiters=osocp.MAXITERS):
    """One instance.  Mk[n,1+m], Bk[1+m,1+m] = posterior of the learned residual at x (0 and I for the fixed-kernel
    model), A[n,n].  Returns dict(u[2], relax, status, iterations, cones, terms, x_next[3], sol)."""
    x = np.asarray(x, dtype=np.float64)
    grad, const, sign = constraint_rows(x, plan, dot_plan, Kp, clf_gamma, centers, radii, tw, gammas)
    K = len(sign)
    if relax_mask is None:
        relax_mask = np.array([1.0] + [0.0] * (K - 1))
    fhat, ghat = ouni.ackermann_f(x), ouni.ackermann_g(x, L_mean)
    terms = [ocbc.reldeg1_terms(Mk, Bk, A, grad[k], const[k], fhat, ghat, sign=sign[k]) for k in range(K)]
    try:
        cones = [ocbc.convert_cbc_terms_to_socp_terms(*tm, 0) for tm in terms]
    except np.linalg.LinAlgError:      # Asq not positive definite: torch.linalg.cholesky raises in the reference (:861)
        return dict(u=np.full(2, np.nan), relax=np.nan, status="bad_cone", iterations=0, cones=None, terms=terms,
                    x_next=x.copy(), sol=None, grad=grad, const=const)
    sol = osocp.clf_cbf_socp(w, r, cones, rho, relax_mask, maxiters=maxiters)
    ok = sol["status"] == "optimal"
    u = sol["x"][:2]
    # the reference raises ValueError(problem.status) when the program is not solved (:954-964): no step is taken
    x_next = ouni.ackermann_step(x, u, dt, L_true) if (ok and dt > 0) else x.copy()
    return dict(u=u, relax=sol["x"][2], status=sol["status"], iterations=sol["iterations"], cones=cones, terms=terms,
                x_next=x_next, sol=sol, grad=grad, const=const)


def shifted_status(step, w, r, rho, relax_mask, delta):
    """Status of the same program with every un-relaxed cone's offset d_k moved by +-delta (1 + |d_k|): an instance
    whose status flips inside that band is numerically on the feasibility boundary, and a solver working on inputs
    that differ by rounding may legitimately land on either side.  Returns (status_loosened, status_tightened)."""
    out = []
    for sgn in (+1.0, -1.0):
        cones = []
        for (A, b, c, d), rm in zip(step["cones"], relax_mask):
            cones.append((A, b, c, d + (0.0 if rm else sgn * delta * (1.0 + abs(d)))))
        out.append(osocp.clf_cbf_socp(w, r, cones, rho, relax_mask)["status"])
    return tuple(out)
