"""Mirror of bayes_cbf/gp_algebra.py: the GP expression algebra the reference builds its safety conditions with.

The reference evaluates an expression tree (`grad_h.t() @ fu_gp + h_gp`, `GradientGP(L1h).t() @ fu_gp + ...`) node by
node with autograd through `custom_predict` (gp_algebra.py:109-255, 319-402).  Here the same operators build the same
tree, but evaluation LOWERS the tree onto the closed-form device kernels:

    sum_i c_i * Det(g_i).t() @ fu_gp  +  sum_j c_j * Det(s_j)                     -> rel-degree-1 condition
        (bcbf_posterior_query + bcbf_cbc_terms: GaussianProcessAddExpr / DetMatmulExpr / MulExpr rules)
    GradientGP(Det(grad_h).t() @ f_gp).t() @ fu_gp + c0 * Det(h) + c1 * (Det(grad_h).t() @ f_gp)   -> rel-degree 2
        (bcbf_posterior_jets + bcbf_cbc2_terms: MatmulExpr product-of-Gaussians terms, GradientGP derivative kernels)
    GradientGP(Det(grad_h).t() @ f_gp)   mean(x), knl(x, x)                        -> from the jets

where `f_gp` / `fu_gp` are the leaves handed out by a `ControlAffineRegressor` (or a sum of dynamics models around
one).  Any other tree -- and `knl(x, x')` / `covar(Z, x, x')` between two different states -- is evaluated node by node by
the reference's propagation rules in `gp_eval` (jets from the device instead of autograd).  There is no CPU path.
Leaves keep the reference's `mean / knl / covar / register_covar` behaviour (gp_algebra.py:70-106, 258-315)."""
import torch


class GaussianProcessBase:
    def __add__(self, Y):
        return GaussianProcessAddExpr(self, Y)

    def __radd__(self, Y):                       # sum([...]) starts from 0
        if isinstance(Y, (int, float)) and Y == 0:
            return self
        return GaussianProcessAddExpr(Y, self)

    def __mul__(self, a):
        return GaussianProcessMulExpr(self, a)

    __rmul__ = __mul__

    def __truediv__(self, a):
        return GaussianProcessMulExpr(self, 1 / a)

    def __neg__(self):
        return GaussianProcessMulExpr(self, -1.0)

    def __matmul__(self, Y):
        return GaussianProcessMatmulExpr(self, Y)

    def t(self):
        return GaussianProcessTranspose(self)


class DeterministicGP(GaussianProcessBase):
    """A deterministic function seen as a GP with zero covariance (gp_algebra.py:70-106).  `jac` (optional) is the
    analytic Jacobian d mean / dx used by GradientGP instead of autograd on `mean`."""

    def __init__(self, mean, shape, name="{mean}", jac=None):
        self._mean, self._shape, self.jac = mean, shape, jac
        self._name = name.format(mean=mean)

    @property
    def shape(self):
        return self._shape

    def mean(self, x):
        return self._mean(x)

    def knl(self, x, xp):
        k = max(self._shape)
        return x.new_zeros(k, k)

    def covar(self, Z, x, xp):
        return x.new_zeros(max(self._shape), max(Z.shape))


class GaussianProcess(GaussianProcessBase):
    """Leaf GP with registered cross-covariances (gp_algebra.py:258-315).  `source = (model, kind, u)` with kind
    'f' or 'fu' marks the leaves a dynamics model hands out; expressions over them evaluate on the device."""

    def __init__(self, mean, knl, shape, assume_independence=False, name="{mean}", source=None):
        self._mean, self._knl, self._shape = mean, knl, shape
        self._covars = dict()
        self.register_covar(self, self.knl)
        self.assume_independence = assume_independence
        self._name = name.format(mean=mean)
        self.source = source

    @property
    def shape(self):
        return self._shape

    def mean(self, x):
        return self._mean(x)

    def knl(self, x, xp):
        return self._knl(x, xp)

    def covar(self, Z, x, xp):
        if isinstance(Z, GaussianProcess):
            if id(Z) in self._covars:
                return self._covars[id(Z)](x, xp)
            if self.assume_independence:
                return x.new_zeros(max(self.shape), max(Z.shape))
            raise ValueError("No covariance registered among two leaf GaussianProcesses")
        if isinstance(Z, DeterministicGP):
            return x.new_zeros(max(self.shape), max(Z.shape))
        c = Z.covar(self, x, xp)                        # a composed expression: its rule, transposed (gp_algebra.py:301-302)
        return c.t() if c.dim() == 2 else c

    def register_covar(self, gp, covar_func):
        """One function for both directions, as the reference does (gp_algebra.py:306-309)."""
        self._covars[id(gp)] = covar_func
        gp._covars[id(self)] = covar_func


# ------------------------------------------------------------------------------------------------ expression nodes
class GaussianProcessExpr(GaussianProcessBase):
    """Inner node: evaluates by lowering the whole tree (see module docstring)."""

    def _lowered(self):
        if getattr(self, "_low", None) is None:
            self._low = lower(self)
        return self._low

    def quadratic_terms(self, x, u0):
        return self._lowered().quadratic_terms(x, u0)

    def _fast(self):
        """The fused lowering when the tree is one of the safety-condition shapes, else None (general evaluation)."""
        if getattr(self, "_low", None) is None and not getattr(self, "_no_low", False):
            try:
                self._low = lower(self)
            except NotImplementedError:
                self._no_low = True
        return getattr(self, "_low", None)

    def mean(self, x):
        low = self._fast()
        if low is not None:
            return low.mean(x)
        from .gp_eval import eval_mean
        return eval_mean(self, x)

    def knl(self, x, xp):
        low = self._fast()
        if low is not None and (xp is x or torch.equal(x, xp)):
            return low.knl(x, xp)
        from .gp_eval import eval_knl
        return eval_knl(self, x, xp)

    def covar(self, Z, x, xp):
        from .gp_eval import eval_covar
        return eval_covar(self, Z, x, xp)


class GaussianProcessAddExpr(GaussianProcessExpr):
    def __init__(self, X, Y):
        assert isinstance(X, GaussianProcessBase) and isinstance(Y, GaussianProcessBase)
        self.lhs, self.rhs = X, Y

    @property
    def shape(self):
        return self.lhs.shape


class GaussianProcessMulExpr(GaussianProcessExpr):
    def __init__(self, X, a):
        assert isinstance(X, GaussianProcessBase)
        assert isinstance(a, (float, int, torch.Tensor))
        self.rhs, self.a = X, float(a)

    @property
    def shape(self):
        return self.rhs.shape


class GaussianProcessTranspose(GaussianProcessExpr):
    def __init__(self, gp):
        assert isinstance(gp, GaussianProcessBase)
        self.gp = gp

    @property
    def shape(self):
        s = self.gp.shape
        return (s[1],) if len(s) == 2 else (1, s[0])

    def t(self):
        return self.gp

    def mean(self, x):
        return self.gp.mean(x)

    def knl(self, x, xp):
        return self.gp.knl(x, xp)

    def covar(self, Y, x, xp):
        c = self.gp.covar(Y, x, xp)
        return c.t() if c.dim() == 2 else c


class GaussianProcessMatmulExpr(GaussianProcessExpr):
    """X @ Y with X a transposed vector GP: the inner product X'Y, a scalar GP (gp_algebra.py:133-199; the
    reference's DetMatmulExpr is the special case of a deterministic X)."""

    def __init__(self, X, Y):
        assert isinstance(X, GaussianProcessBase) and isinstance(Y, GaussianProcessBase)
        assert X.shape[-1] == Y.shape[0]
        self.lhs = X.t()          # the column vector, as the reference stores it
        self.rhs = Y

    @property
    def shape(self):
        return (1,)


GaussianProcessDetMatmulExpr = GaussianProcessMatmulExpr


EPS = 2e-3            # gp_algebra.py:317

# How the Hessian clean-up of GradientGP.knl (gp_algebra.py:384-392) rebuilds the matrix when an eigenvalue lies in
# (-EPS, 0):  "reference" = the reference's own `eigenvectors.T @ diag(evalz) @ eigenvectors` on the GENERAL eigen-solver's
# output (the default: results identical to the reference's);  "project" = the spectral projection V max(L, 0) V' of the
# symmetric part (what a PSD clean-up is usually meant to be; differs from the reference whenever the branch runs).
HESSIAN_CLEANUP = "reference"


def clean_kernel_hessian(H, eigeps=EPS, mode=None):
    """gp_algebra.py:384-392 on one n x n Hessian (host side: n <= 4).  Returns (H_clean, fired).

    mode "reference" is the reference's code path statement by statement: `torch.eig` (= LAPACK xGEEV; torch.linalg.eig
    today) on H in its own dtype, `assert (eigenvalues > -eigeps).all()`, eigenvalues in (-eigeps, 0) set to zero,
    `eigenvectors.T @ diag(evalz) @ eigenvectors`.  NB that product pairs eigenvalue k with ROW k of the eigenvector
    matrix, so it depends on the solver's eigenvalue order and eigenvector signs (csrc/geev_small.h)."""
    mode = mode or HESSIAN_CLEANUP
    if mode == "project":
        w, V = torch.linalg.eigh((0.5 * (H + H.t())).cpu())
        assert bool((w > -eigeps).all()), " Hessian must be positive definite"
        if bool((w < 0).any()):
            return ((V * w.clamp_min(0.0)) @ V.t()).to(H), True
        return H, False
    if mode != "reference":
        raise ValueError("HESSIAN_CLEANUP must be 'reference' or 'project', got %r" % (mode,))
    w, V = torch.linalg.eig(H.detach().cpu())
    evalz, eigenvectors = w.real.clone(), V.real
    assert bool((evalz > -eigeps).all()), " Hessian must be positive definite"
    small_neg_eig = (evalz > -eigeps) & (evalz < 0)
    if bool(small_neg_eig.any()):
        evalz[small_neg_eig] = 0
        return (eigenvectors.t() @ torch.diag(evalz) @ eigenvectors).to(H), True
    return H, False


class GradientGP(GaussianProcessExpr):
    """grad_x of a scalar GP expression (gp_algebra.py:319-402).  Supported operand: Det(grad_h).t() @ f_gp (the Lie
    derivative L_f h); mean(x) = grad (grad_h' m_f)(x), knl(x, x) = d2/dx dx' of its kernel, from the posterior jets."""

    def __init__(self, f, x_shape, grad_check=False, analytical_hessian=True):
        self.gp, self.x_shape = f, x_shape

    @property
    def shape(self):
        return self.x_shape

    def _lie1(self):
        terms, _ = _flatten(self.gp, 1.0)
        if len(terms) != 1 or terms[0][1][0] != "L1" or terms[0][0] != 1.0 or terms[0][1][2].source[1] != "f":
            raise NotImplementedError("GradientGP is evaluated for Det(grad_h).t() @ f_func_gp() only")
        _, (_, grad_gp, leaf) = terms[0]
        return grad_gp, leaf

    def _lie1_or_none(self):
        try:
            return self._lie1()
        except (NotImplementedError, AttributeError):
            return None

    def mean(self, x):
        from .cbc2 import lie1_gradient
        l1 = self._lie1_or_none()
        if l1 is not None:
            return lie1_gradient(l1[1].source[0], l1[0], x)[0]
        from .gp_eval import eval_mean
        return eval_mean(self, x)

    def knl(self, x, xp):
        from .cbc2 import lie1_gradient
        l1 = self._lie1_or_none()
        if l1 is not None and (xp is x or torch.equal(x, xp)):
            return lie1_gradient(l1[1].source[0], l1[0], x)[1]
        from .gp_eval import eval_knl
        return eval_knl(self, x, xp)


# ------------------------------------------------------------------------------------------------ lowering
def _flatten(e, coef):
    """-> ([(coef, term)], None) with term one of
        ("det", DeterministicGP)                                 scalar deterministic summand
        ("L1", Det(grad), leaf)                                  Det(grad).t() @ leaf
        ("L2", Det(grad), leaf_f, leaf_fu)                       GradientGP(Det(grad).t() @ leaf_f).t() @ leaf_fu"""
    if isinstance(e, GaussianProcessAddExpr):
        return _flatten(e.lhs, coef)[0] + _flatten(e.rhs, coef)[0], None
    if isinstance(e, GaussianProcessMulExpr):
        return _flatten(e.rhs, coef * e.a)[0], None
    if isinstance(e, DeterministicGP):
        return [(coef, ("det", e))], None
    if isinstance(e, GaussianProcessMatmulExpr):
        X, Y = e.lhs, e.rhs
        if isinstance(Y, DeterministicGP) and not isinstance(X, DeterministicGP):
            X, Y = Y, X                                        # inner product is symmetric
        if isinstance(Y, GaussianProcess) and Y.source is not None:
            if isinstance(X, DeterministicGP):
                return [(coef, ("L1", X, Y))], None
            if isinstance(X, GradientGP):
                grad_gp, leaf_f = X._lie1()
                return [(coef, ("L2", grad_gp, leaf_f, Y))], None
    raise NotImplementedError("expression %s has no closed-form lowering (module docstring lists the supported shapes)"
                              % type(e).__name__)


def lower(expr):
    """Expression tree -> cbc2.CBCExpr (rel-degree 1 or 2) evaluated by the device kernels."""
    from .cbc2 import CBCExpr
    terms, _ = _flatten(expr, 1.0)
    dets = [(c, t[1]) for c, t in terms if t[0] == "det"]
    l1s = [(c, t) for c, t in terms if t[0] == "L1"]
    l2s = [(c, t) for c, t in terms if t[0] == "L2"]

    def det_sum(scale=1.0):
        def fn(x):
            out = 0.0
            for c, d in dets:
                out = out + (c / scale) * torch.as_tensor(d.mean(x)).reshape(()).to(x)
            return out if dets else x.new_zeros(())
        return fn

    if not l2s:
        if not l1s:
            raise NotImplementedError("expression has no random term")
        leaf0 = l1s[0][1][2]
        model, kind, u = leaf0.source
        for _, t in l1s[1:]:
            mdl, knd, uu = t[2].source
            if t[2] is not leaf0 and (mdl is not model or knd != kind or uu is not u):
                raise NotImplementedError("all random terms of a rel-degree-1 condition must be one model's f_func_gp() "
                                          "or fu_func_gp(u) for one u")
        if kind == "f":
            u = torch.zeros(model.ctrl_size)

        def grad_sum(x):
            out = 0.0
            for c, t in l1s:
                out = out + c * torch.as_tensor(t[1].mean(x)).to(x)
            return out
        return CBCExpr(1, None, grad_sum, model, u, cst_fn=det_sum())
    if len(l2s) != 1:
        raise NotImplementedError("one second Lie derivative term per condition")
    c2, (_, grad_gp, leaf_f, leaf_fu) = l2s[0]
    model, _, u = leaf_fu.source
    if leaf_f.source[0] is not model:
        raise NotImplementedError("L_f h and f + g u must come from the same model")
    ka1 = 0.0
    for c, t in l1s:
        if t[1] is not grad_gp or t[2].source[1] != "f" or t[2].source[0] is not model:
            raise NotImplementedError("the first-order term of a rel-degree-2 condition must be the same L_f h")
        ka1 += c / c2
    hess = grad_gp.jac
    # CBC2 / c2 = L_f^2 h + 1 * (sum_j c_j s_j / c2) + ka1 * L_f h
    return CBCExpr(2, det_sum(c2), grad_gp.mean, model, u, k_alpha=[1.0, ka1], hess_h=hess, scale=c2)
