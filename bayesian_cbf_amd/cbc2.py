"""Mirror of bayes_cbf/cbc2.py + cbc1.py: safety factors, and `cbc2_quadratic_terms` for
rel-degree-1 conditions in closed form on the device."""
import math

import torch
from scipy.special import erfinv

from . import ops


def cbc1_safety_factor(delta):
    """sqrt(2) erfinv(1 - 2 delta)   (bayes_cbf/cbc1.py:10-14)."""
    assert delta < 0.5
    return math.sqrt(2) * float(erfinv(1 - 2 * delta))


def cbc2_safety_factor(delta):
    """sqrt((1 - delta)/delta)   (bayes_cbf/cbc2.py:36-40)."""
    assert delta < 0.5
    return math.sqrt((1 - delta) / delta)


def reldeg1_quadratic_terms(Mk, Bk, A, grad, cst, sign, fhat, ghat):
    """Batched closed form of `cbc2_quadratic_terms` (cbc2.py:7-23) for conditions
    sign*(grad'(fhat + ghat u + F(x)[1;u]) + cst):  returns ((mean_A[B,K,m], mean_b[B,K]),
    (k_Q[B,K,m,m], k_p[B,K,m], k_r[B,K])) -- the reference's ((bfe, e), (V, bfv, v))."""
    terms, cones, cstatus = ops.cbc_terms(Mk, Bk, A, grad, cst, sign, fhat, ghat)
    bfe, e, V, bfv, v = ops.unpack_terms(terms, ghat.shape[2])
    return (bfe, e), (V, bfv, v)
