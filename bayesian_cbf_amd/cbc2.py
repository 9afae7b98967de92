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


def cbc2_quadratic_terms(regressor, h, grad_h, hess_h, x, u0, k_alpha):
    """Rel-degree-2 counterpart of the reference call
        cbc2_quadratic_terms(lambda u: cbc2_gp(h, grad_h, regressor, u, k_alpha), x, u0)   (cbc2.py:7-33)
    for a `ControlAffineRegressor` façade object.  `hess_h(x)` replaces the autograd pass through
    `grad_h` that GradientGP performs (gp_algebra.py:340-345).  x[n] or [b,n], u0[m] or [b,m].
    Returns ((mean_A, mean_b), (k_Q, k_p, k_r), mean, var) like the reference."""
    single = x.dim() == 1
    xb = regressor._ensure_device_dtype(x.reshape(-1, regressor.x_dim)).contiguous()
    ub = regressor._ensure_device_dtype(u0.reshape(-1, regressor.u_dim)).contiguous()
    b = xb.shape[0]
    st = regressor._state()
    Mk, Bk, G, Mj = ops.posterior_jets(st["Lop"], st["Vw"], st["X"], st["UHB"], st["ell"], st["s2"], st["Bm"],
                                       st["M0"], xb, shared=True)
    f = dict(dtype=xb.dtype, device=xb.device)
    hv = torch.stack([torch.as_tensor(h(xi), **f).reshape(()) for xi in xb])
    gh = torch.stack([torch.as_tensor(grad_h(xi), **f) for xi in xb]).contiguous()
    Hh = torch.stack([torch.as_tensor(hess_h(xi), **f) for xi in xb]).contiguous()
    rep = lambda t: t.expand(b, *t.shape[1:]).contiguous()
    (mA, mb), (Q, p, r), mean, var, status = ops.cbc2_terms(
        Mk, Bk, G, Mj, rep(st["A"]), rep(st["Bm"]), rep(st["ell"]), rep(st["s2"]), hv.contiguous(), gh, Hh,
        torch.as_tensor(k_alpha, **f), ub)
    if bool((status != 0).any()):
        raise AssertionError(" Hessian must be positive definite")      # gp_algebra.py:386
    if single:
        return (mA[0], mb[0]), (Q[0], p[0], r[0]), mean[0], var[0]
    return (mA, mb), (Q, p, r), mean, var
