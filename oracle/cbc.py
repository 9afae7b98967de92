"""Oracle: control-barrier / control-Lyapunov condition terms (test infrastructure).

Closed forms of what the reference obtains by building a `gp_algebra` expression and
differentiating it with autograd (`cbc2_quadratic_terms`, bayes_cbf/cbc2.py:7-23 with
misc.py:268-285): at a state x the mean of a rel-degree-1 condition is affine in u and its
variance is quadratic in u,
    mean(u) = bfe'u + e,        var(u) = u'V u + bfv'u + v.
"""
import math
import numpy as np
from scipy.special import erfinv


def cbc1_safety_factor(delta):
    """sqrt(2) erfinv(1 - 2 delta)  (bayes_cbf/cbc1.py:10-14; ControllerCLFBayesian._factor,
    unicycle_move_to_pose.py:922-924 allows delta = 0.5 -> 0)."""
    return math.sqrt(2.0) * float(erfinv(1.0 - 2.0 * delta))


def cbc2_safety_factor(delta):
    """sqrt((1-delta)/delta)  (bayes_cbf/cbc2.py:36-40)."""
    return math.sqrt((1.0 - delta) / delta)


def reldeg1_terms(Mk, Bk, A, grad_h, const, fhat, ghat, sign=1.0):
    """Rel-degree-1 condition  sign * (grad_h' (fhat + ghat u + F(x)[1;u]) + const)  as a GP in u.

    Follows the expression the reference builds at bayes_cbf/unicycle_move_to_pose.py:901-906
    (`_cbc`: const = gamma*h(x)) and :880-888 (`_clc`: const = grad_goal_V' xdot_plan + gamma*V,
    sign = -1), with F ~ MVGP(Mk, A, Bk) the learned residual and fhat, ghat the deterministic
    prior dynamics (:388-397); gp_algebra propagation rules gp_algebra.py:109-168, 201-223.
    Mk[n,1+m], Bk[1+m,1+m], A[n,n], grad_h[n], fhat[n], ghat[n,m].
    Returns (bfe[m], e, V[m,m], bfv[m], v) exactly as cbc2_quadratic_terms (cbc2.py:7-23).
    """
    g = np.asarray(grad_h, dtype=np.float64)
    bfe = sign * ((ghat + Mk[:, 1:]).T @ g)
    e = sign * (g @ (fhat + Mk[:, 0]) + const)
    a_h = g @ A @ g
    V = a_h * Bk[1:, 1:]
    bfv = 2.0 * a_h * Bk[1:, 0]
    v = a_h * Bk[0, 0]
    return bfe, e, V, bfv, v


def convert_cbc_terms_to_socp_terms(bfe, e, V, bfv, v, extravars=0):
    """(bfe,e,V,bfv,v) -> cone form  c'y + d >= |A y + b|   (unicycle_move_to_pose.py:837-878;
    twin at controllers.py:423-482).  Asq = [[v, bfv'/2],[bfv/2, V]] = L L';  A = L'[:,1:],
    b = L'[:,0], c = [.., 1, bfe] (a 1 on the last extra variable), d = e."""
    m = len(bfe)
    Asq = np.empty((m + 1, m + 1))
    Asq[0, 0] = v
    Asq[0, 1:] = np.asarray(bfv) / 2.0
    Asq[1:, 0] = np.asarray(bfv) / 2.0
    Asq[1:, 1:] = V
    L = np.linalg.cholesky(Asq)
    A = np.zeros((m + 1, m + extravars))
    A[:, extravars:] = L.T[:, 1:]
    bfb = L.T[:, 0].copy()
    bfc = np.zeros(m + extravars)
    if extravars >= 1:
        bfc[extravars - 1] = 1.0
    bfc[extravars:] = bfe
    return A, bfb, bfc, e
