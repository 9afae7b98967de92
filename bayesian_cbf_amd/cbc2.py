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


def resolve_model(model):
    """(regressor, [deterministic models]) of a Bayesian dynamics model: a `ControlAffineRegressor` itself, or a sum of
    models (`controllers.SumDynamicModels` / `MeanAdjustedModel`, controllers.py:288-378) of which exactly one is a
    regressor -- the others contribute f_func / g_func to the mean only (DeterministicGP, gp_algebra.py:70-106)."""
    if hasattr(model, "_state") and hasattr(model, "model"):
        return model, []
    if hasattr(model, "fixed_kernel"):          # deterministic mean + a fixed kernel (AckermannDrive.fu_func_gp,
        return FixedKernelGP(model), [model]     # unicycle_move_to_pose.py:261-275; CartesianDynamics :190-197)
    parts = getattr(model, "models", None)
    if parts is None:
        raise TypeError("model must be a ControlAffineRegressor or a sum of dynamics models, got %r" % type(model))
    regs = [p for p in parts if hasattr(p, "_state")]
    if len(regs) != 1:
        raise TypeError("a summed model needs exactly one learned regressor, found %d" % len(regs))
    return regs[0], [p for p in parts if p is not regs[0]]


class FixedKernelGP:
    """Stand-in for the regressor of a model whose GP is its deterministic mean with a state-independent kernel
    (u_hom' B u_hom) A: "posterior" M_k = 0, B_k = B, no data.  Only rel-degree-1 conditions lower onto it."""
    Xtrain = None

    def __init__(self, model):
        self.owner = model
        self.x_dim, self.u_dim = model.state_size, model.ctrl_size
        self.device, self.dtype = torch.device("cuda"), torch.float64

    def _ensure_device_dtype(self, X):
        return torch.as_tensor(X).to(device=self.device, dtype=self.dtype)

    def _require_gpu(self):
        if not torch.cuda.is_available():
            raise RuntimeError("bayesian_cbf_amd runs its arithmetic in libbcbf on a ROCm GPU; there is no CPU path")

    def _hyper(self):
        A, B = self.owner.fixed_kernel()
        f = dict(device=self.device, dtype=self.dtype)
        n, C = self.x_dim, 1 + self.u_dim
        return dict(A=torch.as_tensor(A).to(**f).reshape(1, n, n).contiguous(), Bm=torch.as_tensor(B).to(**f).reshape(1, C, C).contiguous(),
                    ell=torch.ones(1, n, **f), s2=torch.ones(1, **f), M0=torch.zeros(1, C, n, **f))


def _det_mean(dets, xb, reg, want_jac):
    """fhat[b,n], ghat[b,n,m] (and dfhat/dx [b,n,n] by autograd on the user's deterministic functions)."""
    b, n, m = xb.shape[0], reg.x_dim, reg.u_dim
    fhat, ghat = xb.new_zeros(b, n), xb.new_zeros(b, n, m)
    J = xb.new_zeros(b, n, n) if want_jac else None
    for d in dets:
        fhat = fhat + torch.as_tensor(d.f_func(xb), dtype=xb.dtype, device=xb.device).reshape(b, n)
        ghat = ghat + torch.as_tensor(d.g_func(xb), dtype=xb.dtype, device=xb.device).reshape(b, n, m)
        if want_jac:
            for i in range(b):
                Ji = torch.autograd.functional.jacobian(
                    lambda z: torch.as_tensor(d.f_func(z), dtype=z.dtype, device=z.device).reshape(n), xb[i].clone())
                J[i] += Ji.reshape(n, n)
    return fhat.contiguous(), ghat.contiguous(), J


def posterior_for(reg, xb, jets):
    """(st, Mk, Bk, G, Mj) at the query states; the prior when the regressor holds no data
    (control_affine_model.py:495-506)."""
    b, n, C = xb.shape[0], reg.x_dim, 1 + reg.u_dim
    if reg.Xtrain is None:
        reg._require_gpu()
        hp = reg._hyper()
        Mk = hp["M0"].transpose(1, 2).expand(b, n, C).contiguous()
        Bk = (hp["s2"].reshape(1, 1, 1) * hp["Bm"]).expand(b, C, C).contiguous()
        CT = C * (1 + n)
        return hp, Mk, Bk, xb.new_zeros(b, CT, CT), xb.new_zeros(b, n, CT)
    st = reg._state()
    if not jets:
        _, Mk, Bk, _ = reg._query(xb, want_W=False)
        return st, Mk, Bk, None, None
    Mk, Bk, G, Mj = ops.posterior_jets(st["Lop"], st["Vw"], st["X"], st["UHB"], st["ell"], st["s2"], st["Bm"],
                                       st["M0"], xb, shared=True, kernel=getattr(reg, "data_kernel", "rbf"))
    return st, Mk, Bk, G, Mj


def _hessian_mode():
    from . import gp_algebra
    return gp_algebra.HESSIAN_CLEANUP


def reldeg2_quadratic_terms(regressor, h, grad_h, hess_h, x, u0, k_alpha):
    """Closed form of the reference call
        cbc2_quadratic_terms(lambda u: cbc2_gp(h, grad_h, regressor, u, k_alpha), x, u0)   (cbc2.py:7-33)
    for a `ControlAffineRegressor` façade object.  `hess_h(x)` replaces the autograd pass through
    `grad_h` that GradientGP performs (gp_algebra.py:340-345).  x[n] or [b,n], u0[m] or [b,m].
    Returns ((mean_A, mean_b), (k_Q, k_p, k_r), mean, var) like the reference."""
    single = x.dim() == 1
    regressor, dets = resolve_model(regressor)
    if isinstance(regressor, FixedKernelGP):
        raise NotImplementedError("rel-degree-2 conditions need the derivative kernel of a learned (RBF) model")
    xb = regressor._ensure_device_dtype(x.reshape(-1, regressor.x_dim)).contiguous()
    ub = regressor._ensure_device_dtype(u0.reshape(-1, regressor.u_dim)).contiguous()
    b = xb.shape[0]
    st, Mk, Bk, G, Mj = posterior_for(regressor, xb, jets=True)
    if dets:            # deterministic summands shift the mean of [f g] and of df/dx only
        C = 1 + regressor.u_dim
        fhat, ghat, J = _det_mean(dets, xb, regressor, want_jac=True)
        Mk = Mk + torch.cat([fhat.unsqueeze(-1), ghat], dim=-1)
        Mj = Mj.clone()
        for i in range(regressor.x_dim):
            Mj[:, :, (1 + i) * C] += J[:, :, i]
        Mk, Mj = Mk.contiguous(), Mj.contiguous()
    f = dict(dtype=xb.dtype, device=xb.device)
    hv = torch.stack([torch.as_tensor(h(xi), **f).reshape(()) for xi in xb])
    gh = torch.stack([torch.as_tensor(grad_h(xi), **f) for xi in xb]).contiguous()
    Hh = torch.stack([torch.as_tensor(hess_h(xi), **f) for xi in xb]).contiguous()
    rep = lambda t: t.expand(b, *t.shape[1:]).contiguous()
    (mA, mb), (Q, p, r), mean, var, status = ops.cbc2_terms(
        Mk, Bk, G, Mj, rep(st["A"]), rep(st["Bm"]), rep(st["ell"]), rep(st["s2"]), hv.contiguous(), gh, Hh,
        torch.as_tensor(k_alpha, **f), ub, hessian_mode=_hessian_mode(), kernel=getattr(regressor, "data_kernel", "rbf"))
    if bool((status == 1).any()):
        raise AssertionError(" Hessian must be positive definite")      # gp_algebra.py:386
    if single:
        return (mA[0], mb[0]), (Q[0], p[0], r[0]), mean[0], var[0]
    return (mA, mb), (Q, p, r), mean, var


# ------------------------------------------------------------------------------------------------
# The reference's own call shapes (cbc2.py:7-66, cbc1.py:17-52).  The reference builds a GP expression tree
# (`grad_h.t() @ f_gp`, GradientGP, ...) and differentiates it with autograd; here `cbc2_gp` / `RelDeg*Safety.cbc`
# return a light handle that remembers (h, grad_h, model, u, k_alpha) and evaluates through the closed-form kernels.
def _hessian_of(grad_h, x):
    """d grad_h / dx by autograd on the user's deterministic task function (n <= 8): what GradientGP obtains by
    differentiating through grad_h (gp_algebra.py:340-345)."""
    xr = x.detach().clone().requires_grad_(True)
    return torch.autograd.functional.jacobian(lambda z: torch.as_tensor(grad_h(z), dtype=z.dtype, device=z.device), xr)


class CBCExpr:
    """Value of `cbc(u)`: a scalar GP in x for the fixed control u, with the reference's `.mean(x)` / `.knl(x, x)`.
    `expr * c` (GaussianProcessMulExpr, gp_algebra.py:201-223) scales the mean by c and the variance by c^2 -- the
    reference writes its Lyapunov condition as `cbc * -1.0` (unicycle_move_to_pose.py:880-888)."""

    def __init__(self, rel_degree, h, grad_h, model, u, k_alpha=None, gamma=None, hess_h=None, scale=1.0, cst_fn=None):
        self.rel_degree, self.h, self.grad_h, self.model, self.u = rel_degree, h, grad_h, model, u
        self.k_alpha, self.gamma, self.hess_h, self.scale = k_alpha, gamma, hess_h, float(scale)
        self.cst_fn = cst_fn if cst_fn is not None else (lambda x: self.gamma * torch.as_tensor(self.h(x)))

    def __mul__(self, c):
        return CBCExpr(self.rel_degree, self.h, self.grad_h, self.model, self.u, k_alpha=self.k_alpha, gamma=self.gamma,
                       hess_h=self.hess_h, scale=self.scale * float(c), cst_fn=self.cst_fn)

    __rmul__ = __mul__

    def __neg__(self):
        return self * -1.0

    def _scaled(self, res):
        (mA, mb), (Q, p, r), mean, var = res
        c = self.scale
        if c == 1.0:
            return res
        return (c * mA, c * mb), (c * c * Q, c * c * p, c * c * r), c * mean, c * c * var

    def quadratic_terms(self, x, u0):
        """((mean_A, mean_b), (Q, p, r), mean(u0), var(u0)); x[n] / u0[m], or a batch x[b,n] / u0[b,m]."""
        if self.rel_degree == 2:
            hess = self.hess_h if self.hess_h is not None else (lambda z: _hessian_of(self.grad_h, z))
            return self._scaled(reldeg2_quadratic_terms(self.model, self.h, self.grad_h, hess, x, u0, self.k_alpha))
        reg, dets = resolve_model(self.model)
        single = x.dim() == 1
        xb = reg._ensure_device_dtype(x.reshape(-1, reg.x_dim)).contiguous()
        ub = reg._ensure_device_dtype(u0.reshape(-1, reg.u_dim))
        b, n, m = xb.shape[0], reg.x_dim, reg.u_dim
        st, Mk, Bk, _, _ = posterior_for(reg, xb, jets=False)
        f = dict(dtype=xb.dtype, device=xb.device)
        grad = torch.stack([torch.as_tensor(self.grad_h(xi), **f).reshape(n) for xi in xb]).reshape(b, 1, n).contiguous()
        cst = torch.stack([torch.as_tensor(self.cst_fn(xi), **f).reshape(()) for xi in xb]).reshape(b, 1).contiguous()
        fhat, ghat, _ = _det_mean(dets, xb, reg, want_jac=False)
        A = st["A"].expand(b, n, n).contiguous()
        (bfe, e), (V, bfv, v) = reldeg1_quadratic_terms(Mk, Bk, A, grad, cst, torch.ones(1, **f), fhat, ghat)
        mA, mb, Q, p, r = bfe[:, 0], e[:, 0], V[:, 0], bfv[:, 0], v[:, 0]
        mean = (mA * ub).sum(-1) + mb
        var = torch.einsum("bi,bij,bj->b", ub, Q, ub) + (p * ub).sum(-1) + r
        if single:
            return self._scaled(((mA[0], mb[0]), (Q[0], p[0], r[0]), mean[0], var[0]))
        return self._scaled(((mA, mb), (Q, p, r), mean, var))

    def mean(self, x):
        return self.quadratic_terms(x, self.u)[2]

    def knl(self, x, xp):
        if xp is not x and not torch.equal(x, xp):
            raise NotImplementedError("cross-covariance of a constraint between two states is not on the hot path")
        return self.quadratic_terms(x, self.u)[3]


def pack_terms(res):
    """((mean_A, mean_b), (Q, p, r), ..) -> the packed row [.., T] of bcbf_cbc_terms / bcbf_controller_cones."""
    (mA, mb), (Q, p, r) = res[0], res[1]
    lead = mA.shape[:-1]
    return torch.cat([mA, mb.reshape(*lead, 1), Q.reshape(*lead, -1), p, r.reshape(*lead, 1)], dim=-1)


def lie1_gradient(model, grad_gp, x, eigeps=2e-3):
    """GradientGP(Det(grad_h).t() @ f_gp) at x: (grad of the mean [n], d2 k / dx dx' at x = x' [n,n]) from the posterior
    jets (gp_algebra.py:340-402), with the reference's eigenvalue check (> -2e-3) and clean-up of the Hessian."""
    reg, dets = resolve_model(model)
    xb = reg._ensure_device_dtype(x.reshape(1, -1)).contiguous()
    n, C = reg.x_dim, 1 + reg.u_dim
    st, Mk, Bk, G, Mj = posterior_for(reg, xb, jets=True)
    f = dict(dtype=xb.dtype, device=xb.device)
    m0, dm0 = Mk[0, :, 0], torch.stack([Mj[0, :, (1 + i) * C] for i in range(n)], dim=1)     # dm0[j, i] = d m0_j / dx_i
    if dets:
        fhat, _, J = _det_mean(dets, xb, reg, want_jac=True)
        m0, dm0 = m0 + fhat[0], dm0 + J[0]
    gh = torch.as_tensor(grad_gp.mean(xb[0]), **f).reshape(n)
    Hh = torch.as_tensor(grad_gp.jac(xb[0]) if grad_gp.jac is not None else _hessian_of(grad_gp.mean, xb[0]), **f)
    A, B00, ell, s2 = st["A"][0], st["Bm"][0, 0, 0], st["ell"][0], st["s2"][0]
    gmean = Hh.t() @ m0 + dm0.t() @ gh
    idx = [(1 + i) * C for i in range(n)]
    Gm = G[0]
    s00, s_i = Bk[0, 0, 0], -Gm[idx, 0]
    from .data_kernels import kxx as _kxx
    kxx = _kxx(getattr(reg, "data_kernel", "rbf"))     # d2 k / dx dx' at x' = x, in units of s2 / ell^2 (1 | 5/3 | 8/3)
    sij = torch.diag(kxx * s2 / (ell * ell) * B00) - Gm[idx][:, idx]
    Agh = A @ gh
    HAg = Hh @ Agh
    H = (Hh @ A @ Hh) * s00 + torch.outer(HAg, s_i) + torch.outer(s_i, HAg) + (gh @ Agh) * sij
    from .gp_algebra import clean_kernel_hessian
    H, _ = clean_kernel_hessian(H, eigeps)                       # n x n, n <= 4: host side (gp_algebra.py:384-392)
    return gmean.to(dtype=x.dtype, device=x.device), H.to(dtype=x.dtype, device=x.device)


def cbc2_gp(h, grad_h, learned_model, utest, k_alpha, hess_h=None):
    """cbc2.py:26-33, written as the reference writes it; the expression lowers onto the jet kernel + closed-form
    terms (gp_algebra.lower).  `hess_h` (optional) is the analytic d grad_h / dx; autograd on grad_h otherwise."""
    from .gp_algebra import DeterministicGP, GradientGP
    f_gp = learned_model.f_func_gp()
    fu_gp = learned_model.fu_func_gp(utest)
    h_gp = DeterministicGP(h, shape=(1,), name="h(x)")
    grad_h_gp = DeterministicGP(grad_h, shape=(learned_model.state_size,), name="grad h(x)", jac=hess_h)
    L1h = grad_h_gp.t() @ f_gp
    L2h = GradientGP(L1h, x_shape=(learned_model.state_size,)).t() @ fu_gp
    return L2h + h_gp * k_alpha[0] + L1h * k_alpha[1]


def cbc2_quadratic_terms(cbc2, x, u, *more):
    """cbc2.py:7-23 -- `cbc2` is the reference's callable `u -> GP` (e.g. `safety.cbc`); returns
    ((mean_A, mean_b), (k_Q, k_p, k_r), mean(u), var(u)).  (The explicit closed-form signature
    `(regressor, h, grad_h, hess_h, x, u0, k_alpha)` is `reldeg2_quadratic_terms`; it is still accepted here.)"""
    if more:
        return reldeg2_quadratic_terms(cbc2, x, u, *more)
    expr = cbc2(u)
    if not hasattr(expr, "quadratic_terms"):
        # a foreign callable (anything with .mean(x) / .knl(x, x') differentiable in u): the reference's own autograd
        # extraction, on the host (misc.get_affine_terms / get_quadratic_terms)
        from .misc import quadratic_terms_by_autograd
        return quadratic_terms_by_autograd(cbc2, x, u)
    return expr.quadratic_terms(x, u)


class RelDeg2Safety:
    """cbc2.py:42-66: subclasses provide k_alpha, model, max_unsafe_prob, cbf(x), grad_cbf(x)."""

    def cbc(self, u0):
        return cbc2_gp(self.cbf, self.grad_cbf, self.model, u0, self.k_alpha, hess_h=getattr(self, "hess_cbf", None))

    def safety_factor(self):
        return cbc2_safety_factor(self.max_unsafe_prob)
