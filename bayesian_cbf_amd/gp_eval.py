"""General evaluation of `gp_algebra` expression trees (bayes_cbf/gp_algebra.py:109-255, 319-402).

`gp_algebra.lower` recognises the trees of the safety conditions and evaluates them with the fused kernels; everything
else -- any sum / scalar multiple / inner product / transpose / gradient over model leaves and deterministic functions,
`mean(x)`, `knl(x, x')` and `covar(Z, x, x')` at two DIFFERENT states -- is evaluated here, node by node, by the
propagation rules of the reference:

    sum          m = m_X + m_Y,  k = k_X + k_Y + c_YX + c_XY,  c(Z) = c_X(Z) + c_Y(Z)
    a X          m = a m_X,      k = a^2 k_X,                   c(Z) = a c_X(Z)
    X' Y         m = m_X'm_Y + 1/2 tr c_XY(x,x) + 1/2 tr c_YX(x,x)
                 k = 2 tr(c_XY)^2 + m_Y(x)' k_X m_Y(x') + m_X(x)' k_Y m_X(x') + 2 m_Y(x)' c_YX m_X(x')
                 c(Z) = m_X(x)' c_Y(Z) + m_Y(x)' c_X(Z)
    grad X       m = grad_x m_X,  k = d^2 k_X / dx dx',  c(Z) = (d c_X(Z) / dx)'   (both arguments move when x' IS x)

The reference obtains the gradients by autograd through `custom_predict`.  Here every quantity is carried as a JET --
value, first derivatives in x and in x', mixed second derivative -- and the rules are applied to jets (product rule);
the jets of the model leaves come from the device (`bcbf_posterior_jets`: L^-1 [Phi, dPhi/dx_d] at both states, one
launch), those of the deterministic task functions from their analytic Jacobian or torch.autograd on the user's function.
One level of `GradientGP` is supported (rel-degree 2, as upstream uses it); a gradient of a gradient raises.
"""
import torch

from . import ops


# ------------------------------------------------------------------------------------------------ jets
class Jet:
    """v(x, x') [r, c] with dx[i] = dv/dx_i [n, r, c], dp[j] = dv/dx'_j [n, r, c], dxp[i, j] = d2v/dx_i dx'_j
    [n, n, r, c]; None = identically zero."""
    __slots__ = ("v", "dx", "dp", "dxp")

    def __init__(self, v, dx=None, dp=None, dxp=None):
        self.v, self.dx, self.dp, self.dxp = v, dx, dp, dxp


def _add(a, b):
    return b if a is None else a if b is None else a + b


def jadd(a, b):
    return Jet(a.v + b.v, _add(a.dx, b.dx), _add(a.dp, b.dp), _add(a.dxp, b.dxp))


def jscale(a, s):
    f = lambda t: None if t is None else t * s
    return Jet(a.v * s, f(a.dx), f(a.dp), f(a.dxp))


def jT(a):
    f = lambda t: None if t is None else t.transpose(-1, -2)
    return Jet(f(a.v), f(a.dx), f(a.dp), f(a.dxp))


def _mm(a, b):
    return None if a is None or b is None else a @ b


def jmatmul(a, b):
    """Product rule up to the mixed second derivative."""
    dxp = _add(_add(_mm(a.dxp, b.v), _mm(a.v, b.dxp)),
               _add(None if a.dx is None or b.dp is None else torch.einsum("irk,jkc->ijrc", a.dx, b.dp),
                    None if a.dp is None or b.dx is None else torch.einsum("jrk,ikc->ijrc", a.dp, b.dx)))
    return Jet(a.v @ b.v, _add(_mm(a.dx, b.v), _mm(a.v, b.dx)), _add(_mm(a.dp, b.v), _mm(a.v, b.dp)), dxp)


def jtrace(a):
    f = lambda t: None if t is None else t.diagonal(dim1=-2, dim2=-1).sum(-1)[..., None, None]
    return Jet(f(a.v), f(a.dx), f(a.dp), f(a.dxp))


def jdiag(a, var):
    """g(x, x) as a function of ONE variable (first order): the derivative is dx + dp, stored in slot `var`."""
    d = _add(a.dx, a.dp)
    return Jet(a.v, d, None) if var == "x" else Jet(a.v, None, d)


def jzeros(r, c, like):
    return Jet(like.new_zeros(r, c))


# ------------------------------------------------------------------------------------------------ model leaves
def _as_col(t, like):
    return torch.as_tensor(t).to(like).reshape(-1, 1)


class _ModelJets:
    """Posterior of one regressor at the pair (x, x'): M_k and its x-derivatives at both states and the blocks of
    B_k(x, x') = k(x,x') B - W(x)'W(x') with its first and mixed second derivatives, from ONE `bcbf_posterior_jets`
    launch (both states are queries of the shared model; Wj = L^-1 [Phi, dPhi/dx_d])."""

    def __init__(self, model, x, xp, order):
        from .cbc2 import resolve_model, FixedKernelGP, _det_mean
        reg, dets = resolve_model(model)
        self.reg, self.n, self.C = reg, reg.x_dim, 1 + reg.u_dim
        n, C = self.n, self.C
        pts = reg._ensure_device_dtype(torch.stack([x.reshape(-1), xp.reshape(-1)])).contiguous()
        f = dict(dtype=pts.dtype, device=pts.device)
        self.f = f
        fixed = isinstance(reg, FixedKernelGP)
        kernel = "rbf" if fixed else getattr(reg, "data_kernel", "rbf")
        hp = reg._hyper() if (fixed or reg.Xtrain is None) else reg._state()
        self.A = hp["A"][0]
        B, ell, s2 = hp["Bm"][0], hp["ell"][0], hp["s2"][0]
        # ---- prior kernel and its derivatives (RBF-ARD, or the opt-in Matern-5/2; the fixed-kernel models have a constant kernel)
        d = (pts[0] - pts[1]) / (ell * ell)                          # (x - x') / ell^2
        if fixed:
            k, kx, kp, kxp = s2, pts.new_zeros(n), pts.new_zeros(n), pts.new_zeros(n, n)
        else:
            # k = s2 shape(d2);  dk/dx = -s2 dshape d;  d2k / dx_d dx'_e = s2 (dshape delta_de / ell_d^2 + ddshape d_d d_e)
            # (data_kernels.shape_terms: RBF, or the opt-in Matern-5/2 / RBF x Matern-5/2)
            from .data_kernels import shape_terms
            sh, dsh, ddsh = shape_terms(kernel, ((pts[0] - pts[1]) ** 2 / (ell * ell)).sum())
            k = s2 * sh
            kx, kp = -d * (s2 * dsh), d * (s2 * dsh)
            kxp = s2 * (torch.diag(1.0 / (ell * ell)) * dsh + ddsh * torch.outer(d, d))
        self.B00 = k * B
        self.Bx = kx[:, None, None] * B
        self.Bp = kp[:, None, None] * B
        self.Bxp = kxp[:, :, None, None] * B
        self.Mk = hp["M0"][0].t().expand(2, n, C).clone() if "M0" in hp else pts.new_zeros(2, n, C)
        self.dMk = pts.new_zeros(2, n, n, C)                          # [point, d, state, control]
        if not fixed and reg.Xtrain is not None:
            if getattr(reg, "_jets_unsupported", False) or "lin" in hp:
                raise NotImplementedError("derivative jets are built for the RBF data kernel of ControlAffineRegressor")
            st = hp
            if order == 0:
                Mk, _, W = ops.posterior_query(st["Lop"], st["Vw"], st["X"], st["UHB"], st["ell"], st["s2"], st["Bm"],
                                               st["M0"], pts, shared=True, want_W=True, kernel=kernel)
                self.Mk = Mk
                self.B00 = self.B00 - W[0].t() @ W[1]
            else:
                Mk, _, _, Mj, Wj = ops.posterior_jets(st["Lop"], st["Vw"], st["X"], st["UHB"], st["ell"], st["s2"],
                                                      st["Bm"], st["M0"], pts, shared=True, want_W=True, kernel=kernel)
                self.Mk = Mk
                self.dMk = torch.stack([Mj[:, :, (1 + dd) * C:(2 + dd) * C] for dd in range(n)], dim=1)
                G = (Wj[0].t() @ Wj[1]).reshape(1 + n, C, 1 + n, C)    # [a, c, b, c'] = W_a(x)' W_b(x')
                self.B00 = self.B00 - G[0, :, 0, :]
                self.Bx = self.Bx - G[1:, :, 0, :]
                self.Bp = self.Bp - G[0, :, 1:, :].permute(1, 0, 2)
                self.Bxp = self.Bxp - G[1:, :, 1:, :].permute(0, 2, 1, 3)
        # ---- deterministic summands of a summed model (SumDynamicModels): they move the mean only
        if dets:
            fhat, ghat, J = _det_mean(dets, pts, reg, want_jac=order > 0)
            self.Mk = self.Mk + torch.cat([fhat.unsqueeze(-1), ghat], dim=-1)
            if order > 0:
                self.dMk = self.dMk.clone()
                self.dMk[:, :, :, 0] += J.permute(0, 2, 1)                # d fhat_s / dx_d
                for i in range(2):                                        # d ghat / dx by autograd on the user's g_func
                    for dmodel in dets:
                        Jg = torch.autograd.functional.jacobian(
                            lambda z: torch.as_tensor(dmodel.g_func(z), dtype=z.dtype, device=z.device).reshape(n, C - 1),
                            pts[i].clone())                               # [n, m, n_d]
                        self.dMk[i, :, :, 1:] += Jg.permute(2, 0, 1)

    def uh(self, kind, u):
        a = torch.zeros(self.C, **self.f)
        a[0] = 1.0
        if kind == "fu":
            a[1:] = torch.as_tensor(u).to(**self.f).reshape(-1)
        return a

    def mean(self, kind, u, which, var, order):
        a = self.uh(kind, u)
        v = (self.Mk[which] @ a).reshape(-1, 1)
        d = (self.dMk[which] @ a)[:, :, None] if order > 0 else None
        return Jet(v, d, None) if var == "x" else Jet(v, None, d)

    def cov(self, a, ap, order):
        """cov(F(x) a, F(x') a') = (a' B_k(x,x') a') A as a jet."""
        s = lambda Bm: torch.einsum("c,...cd,d->...", a, Bm, ap)
        A = self.A
        if order == 0:
            return Jet(s(self.B00) * A)
        return Jet(s(self.B00) * A, s(self.Bx)[:, None, None] * A, s(self.Bp)[:, None, None] * A,
                   s(self.Bxp)[:, :, None, None] * A if order > 1 else None)


# ------------------------------------------------------------------------------------------------ the evaluator
class Evaluator:
    def __init__(self, order=0):
        self.order = order
        self._mj = {}

    # -- helpers
    def _jets_of(self, model, x, xp):
        key = (id(model), tuple(x.reshape(-1).tolist()), tuple(xp.reshape(-1).tolist()))
        if key not in self._mj:
            self._mj[key] = _ModelJets(model, x, xp, self.order)
        return self._mj[key]

    def _autograd_jet(self, fn, args, rows, cols, like, slots):
        """Jet of a user torch function fn(*args) -> [rows*cols] by torch.autograd (host-side: task functions and hand-made
        leaves).  slots: which of ('x', 'p') each argument is."""
        v = torch.as_tensor(fn(*args)).to(like).reshape(rows, cols)
        out = Jet(v)
        if self.order == 0:
            return out
        n = args[0].numel()
        flat = lambda *a: torch.as_tensor(fn(*a)).to(like).reshape(-1)
        J = torch.autograd.functional.jacobian(flat, tuple(a.detach().clone() for a in args))
        for Ji, slot in zip(J, slots):
            d = Ji.reshape(rows, cols, n).permute(2, 0, 1).to(like)
            if slot == "x":
                out.dx = _add(out.dx, d)
            else:
                out.dp = _add(out.dp, d)
        if self.order > 1 and len(args) == 2:
            def first(xa, xb):
                return torch.autograd.functional.jacobian(lambda z: flat(z, xb), xa, create_graph=True)
            H = torch.autograd.functional.jacobian(lambda xb: first(args[0].detach().clone(), xb), args[1].detach().clone())
            out.dxp = H.reshape(rows, cols, n, n).permute(2, 3, 0, 1).to(like)
        return out

    # -- mean of node e as a function of the variable `var` ('x' or 'p'), at the point pt
    def mean(self, e, pt, var):
        from . import gp_algebra as ga
        if isinstance(e, ga.DeterministicGP):
            k = max(e.shape)
            v = _as_col(e.mean(pt), pt)
            if self.order == 0:
                return Jet(v)
            if e.jac is not None:
                J = torch.as_tensor(e.jac(pt)).to(pt).reshape(k, -1)              # [k, n]
            else:
                J = torch.autograd.functional.jacobian(lambda z: torch.as_tensor(e.mean(z)).to(z).reshape(-1), pt.detach().clone())
            d = J.reshape(k, -1).t()[:, :, None]
            return Jet(v, d, None) if var == "x" else Jet(v, None, d)
        if isinstance(e, ga.GaussianProcess):
            if e.source is not None:
                model, kind, u = e.source
                mj = self._jets_of(model, pt, pt)
                return mj.mean(kind, u, 0, var, self.order)
            return self._autograd_jet(e.mean, (pt,), max(e.shape), 1, pt, (var,))
        if isinstance(e, ga.GaussianProcessAddExpr):
            return jadd(self.mean(e.lhs, pt, var), self.mean(e.rhs, pt, var))
        if isinstance(e, ga.GaussianProcessMulExpr):
            return jscale(self.mean(e.rhs, pt, var), e.a)
        if isinstance(e, ga.GaussianProcessTranspose):
            return jT(self.mean(e.gp, pt, var))
        if isinstance(e, ga.GaussianProcessMatmulExpr):
            X, Y = e.lhs, e.rhs
            m = jmatmul(jT(self.mean(X, pt, var)), self.mean(Y, pt, var))
            cxy = jtrace(self.covar(X, Y, pt, pt, True))
            cyx = jtrace(self.covar(Y, X, pt, pt, True))
            return jadd(m, jscale(jdiag(jadd(cxy, cyx), var), 0.5))
        if isinstance(e, ga.GradientGP):
            if self.order > 0:
                raise NotImplementedError("a gradient of a gradient needs second-order jets (not built: rel-degree <= 2)")
            sub = Evaluator(1)
            m = sub.mean(e.gp, pt, "x")
            n = pt.numel()
            return Jet((m.dx if m.dx is not None else pt.new_zeros(n, 1, 1)).reshape(n, 1))
        if hasattr(e, "mean"):
            return Jet(_as_col(e.mean(pt), pt))
        raise TypeError("not a GP expression: %r" % (e,))

    # -- covariance function of node e between x and x'
    def knl(self, e, x, xp, same):
        from . import gp_algebra as ga
        if isinstance(e, ga.DeterministicGP):
            k = max(e.shape)
            return jzeros(k, k, x)
        if isinstance(e, ga.GaussianProcess):
            return self.covar(e, e, x, xp, same)
        if isinstance(e, ga.GaussianProcessAddExpr):
            X, Y = e.lhs, e.rhs
            return jadd(jadd(self.knl(X, x, xp, same), self.knl(Y, x, xp, same)),
                        jadd(self.covar(Y, X, x, xp, same), self.covar(X, Y, x, xp, same)))
        if isinstance(e, ga.GaussianProcessMulExpr):
            return jscale(self.knl(e.rhs, x, xp, same), e.a * e.a)
        if isinstance(e, ga.GaussianProcessTranspose):
            return self.knl(e.gp, x, xp, same)
        if isinstance(e, ga.GaussianProcessMatmulExpr):
            X, Y = e.lhs, e.rhs
            mXx, mYx = self.mean(X, x, "x"), self.mean(Y, x, "x")
            mXp, mYp = self.mean(X, xp, "p"), self.mean(Y, xp, "p")
            t = jtrace(self.covar(X, Y, x, xp, same))
            out = jscale(jmatmul(t, t), 2.0)
            out = jadd(out, jmatmul(jmatmul(jT(mYx), self.knl(X, x, xp, same)), mYp))
            out = jadd(out, jmatmul(jmatmul(jT(mXx), self.knl(Y, x, xp, same)), mXp))
            return jadd(out, jscale(jmatmul(jmatmul(jT(mYx), self.covar(Y, X, x, xp, same)), mXp), 2.0))
        if isinstance(e, ga.GradientGP):
            if self.order > 0:
                raise NotImplementedError("a gradient of a gradient needs higher-order jets (not built: rel-degree <= 2)")
            sub = Evaluator(2)
            kj = sub.knl(e.gp, x, xp, same)
            n = x.numel()
            H = kj.dxp.reshape(n, n) if kj.dxp is not None else x.new_zeros(n, n)
            if torch.allclose(x, xp):
                H = _clean_hessian(H)
            return Jet(H)
        raise TypeError("not a GP expression: %r" % (e,))

    # -- cov(e(x), Z(x'))  [k_e, k_Z]
    def covar(self, e, Z, x, xp, same):
        from . import gp_algebra as ga
        ke, kz = max(e.shape), max(Z.shape)
        if isinstance(e, ga.DeterministicGP) or isinstance(Z, ga.DeterministicGP):
            return jzeros(ke, kz, x)
        if isinstance(e, ga.GaussianProcess):
            if isinstance(Z, ga.GaussianProcess):
                if id(Z) not in e._covars:
                    if e.assume_independence:
                        return jzeros(ke, kz, x)
                    raise ValueError("No covariance registered among two leaf GaussianProcesses")
                if e.source is not None and Z.source is not None and e.source[0] is Z.source[0]:
                    # one model: cov(F(x) a, F(x') a').  A registered cross-covariance is ONE function for both
                    # directions (gp_algebra.py:306-309): covar_fu_f puts [1; u] on the first argument whichever leaf asks
                    mj = self._jets_of(e.source[0], x, xp)
                    if Z is e:
                        a = ap = mj.uh(e.source[1], e.source[2])
                    else:
                        fu = e if e.source[1] == "fu" else Z
                        a, ap = mj.uh("fu", fu.source[2]), mj.uh("f", None)
                    return mj.cov(a, ap, self.order)
                fn = e._covars[id(Z)]
                return self._autograd_jet(fn, (x, xp), ke, kz, x, ("x", "p"))
            return jT(self.covar(Z, e, x, xp, same))           # as upstream: same argument order, transposed
        if isinstance(e, ga.GaussianProcessAddExpr):
            return jadd(self.covar(e.lhs, Z, x, xp, same), self.covar(e.rhs, Z, x, xp, same))
        if isinstance(e, ga.GaussianProcessMulExpr):
            return jscale(self.covar(e.rhs, Z, x, xp, same), e.a)
        if isinstance(e, ga.GaussianProcessTranspose):
            return self.covar(e.gp, Z, x, xp, same)
        if isinstance(e, ga.GaussianProcessMatmulExpr):
            X, Y = e.lhs, e.rhs
            return jadd(jmatmul(jT(self.mean(X, x, "x")), self.covar(Y, Z, x, xp, same)),
                        jmatmul(jT(self.mean(Y, x, "x")), self.covar(X, Z, x, xp, same)))
        if isinstance(e, ga.GradientGP):
            if self.order > 0:
                raise NotImplementedError("a gradient of a gradient needs higher-order jets (not built: rel-degree <= 2)")
            sub = Evaluator(1)
            c = sub.covar(e.gp, Z, x, xp, same)               # [1, kz]
            n = x.numel()
            d = _add(c.dx, c.dp) if same else c.dx
            return Jet(d.reshape(n, kz) if d is not None else x.new_zeros(n, kz))
        raise TypeError("not a GP expression: %r" % (e,))


def _clean_hessian(H, eigeps=2e-3):
    """gp_algebra.py:384-392 (the reference's formula by default; `gp_algebra.HESSIAN_CLEANUP`)."""
    from .gp_algebra import clean_kernel_hessian
    return clean_kernel_hessian(H, eigeps)[0]


# ------------------------------------------------------------------------------------------------ public entry points
def _shape_out(j, rows_scalar, cols_scalar, like):
    v = j.v
    if rows_scalar and cols_scalar:
        v = v.reshape(())
    elif rows_scalar:
        v = v.reshape(-1)
    elif cols_scalar:
        v = v.reshape(-1)
    return v.to(dtype=like.dtype, device=like.device)


def eval_mean(e, x):
    j = Evaluator(0).mean(e, x, "x")
    scalar = max(e.shape) == 1
    return _shape_out(j, scalar, True, x)


def eval_knl(e, x, xp):
    j = Evaluator(0).knl(e, x, xp, xp is x)
    scalar = max(e.shape) == 1
    return _shape_out(j, scalar, scalar, x)


def eval_covar(e, Z, x, xp):
    j = Evaluator(0).covar(e, Z, x, xp, xp is x)
    return _shape_out(j, max(e.shape) == 1, max(Z.shape) == 1, x)
