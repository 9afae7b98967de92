"""Host helper surface of bayes_cbf/misc.py: the autograd utilities the reference's callers import by
name, the dynamics-model base classes, and the logging names (re-exported from `tblog`).

These are plain torch-autograd host functions on n x n / m x m sized tensors -- not part of the device path.  Inside
this package the closed forms of `cbc2.py` / `gp_eval.py` replace them; they are kept (a) so that code written against
the reference imports unchanged and (b) as the GENERIC fallback of `cbc2_quadratic_terms` for a callable `u -> GP` that
is not one of this package's expression trees (any object with `.mean(x)` and `.knl(x, x')` differentiable in `u`).

Reference lines: t_hstack/t_vstack :29-37, to_numpy :40-44, t_jac :47-53, store_args :56-77, torch_kron :80-106,
DynamicsModel / BayesianDynamicsModel / ZeroDynamicsModel :109-213, isleaf / variable_required_grad :216-233,
t_hessian :236-245, gradgradcheck :248-260, epsilon :262-266, get_affine_terms :269-275, get_quadratic_terms :278-287,
clip :289-290, random_psd :315-317, normalize_radians :320-321, gitdescribe :338-341, ensuredirs :407-411.
"""
import functools
import inspect
import math
import os
import subprocess
from abc import ABC, abstractmethod
from contextlib import contextmanager

import torch

from .tblog import (Logger, NoLogger, TBLogger, load_tensorboard_scalars,  # noqa: F401  (the reference keeps them here)
                    stream_tensorboard_scalars)


def t_hstack(tensors):
    """np.hstack for tensors: concatenate along the last axis."""
    return torch.cat(tensors, dim=-1)


def t_vstack(tensors):
    """np.vstack for tensors: concatenate along the second-to-last axis."""
    return torch.cat(tensors, dim=-2)


def to_numpy(x):
    return x.detach().cpu().double().numpy() if torch.is_tensor(x) else x


def t_jac(f_x, x, retain_graph=False, **kw):
    """d f_x / d x by one reverse sweep per component of f_x (rows of the Jacobian); a 0-d f_x gives the gradient.
    Keyword arguments go to torch.autograd.grad (create_graph, allow_unused ...)."""
    if f_x.ndim == 0:
        return torch.autograd.grad(f_x, x, retain_graph=retain_graph, **kw)[0]
    rows = [torch.autograd.grad(comp, x, retain_graph=True, **kw)[0] for comp in f_x]
    return torch.stack(rows, dim=0)


def store_args(method, skip=()):
    """Decorator for `__init__`-like methods: every parameter (defaults, positionals, keywords; except those named in
    `skip`) is also stored as an attribute of the same name before the method runs."""
    sig = inspect.signature(method)
    names = list(sig.parameters)

    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        for name in names[1:]:
            par = sig.parameters[name]
            if name in skip or par.kind is par.VAR_POSITIONAL:
                continue
            if par.kind is par.VAR_KEYWORD:
                for k, v in bound.arguments.get(name, {}).items():
                    if k not in skip:
                        setattr(self, k, v)
            elif name in bound.arguments:
                setattr(self, name, bound.arguments[name])
        method(self, *args, **kwargs)

    return wrapped


def torch_kron(A, B, batch_dims=1):
    """Kronecker product over the trailing axes, the leading `batch_dims` axes being batch axes (broadcast against each
    other): out[..., i*p + k, j*q + l] = A[..., i, j] * B[..., k, l]."""
    assert A.ndim == B.ndim
    ta, tb = A.shape[batch_dims:], B.shape[batch_dims:]
    Ae = A.reshape(*A.shape[:batch_dims], *[s for d in ta for s in (d, 1)])
    Be = B.reshape(*B.shape[:batch_dims], *[s for d in tb for s in (1, d)])
    prod = Ae * Be
    return prod.reshape(*prod.shape[:batch_dims], *[da * db for da, db in zip(ta, tb)])


class DynamicsModel(ABC):
    """A control-affine plant  xdot = f(x) + g(x) u  with an explicit-Euler `step`."""

    def __init__(self):
        self._state = None

    @property
    @abstractmethod
    def ctrl_size(self):
        """dimension of u"""

    @property
    @abstractmethod
    def state_size(self):
        """dimension of x"""

    @abstractmethod
    def f_func(self, X):
        """f(X) for X [state_size] or [b, state_size]"""

    @abstractmethod
    def g_func(self, X):
        """g(X): [..., state_size, ctrl_size]"""

    def normalize_state(self, X_in):
        return X_in

    def forward(self, x, u):
        single = x.ndim == 1
        Xb = x[None] if single else x
        Ub = {1: lambda: u[None, :, None], 2: lambda: u[None]}.get(u.ndim, lambda: u)()
        Xdot = self.f_func(Xb) + torch.bmm(self.g_func(Xb), Ub).squeeze(-1)
        return Xdot[0] if single else Xdot

    def step(self, u, dt):
        xdot = self.forward(self._state, u)
        self._state = self.normalize_state(self._state + xdot * dt)
        return dict(x=self._state, xdot=xdot)

    def set_init_state(self, x0):
        self._state = x0.clone()

    def F_func(self, X):
        """[f(X), g(X)] side by side: [..., state_size, 1 + ctrl_size]"""
        return torch.cat([self.f_func(X).unsqueeze(-1), self.g_func(X)], dim=-1)


class BayesianDynamicsModel(DynamicsModel):
    @abstractmethod
    def fu_func_gp(self, U):
        """the GP of f(x) + g(x) U as a function of x (gp_algebra expression)"""


class ZeroDynamicsModel(DynamicsModel):
    """f = 0, g = 0 (kept differentiable in X, as the reference's products with X are)."""

    def __init__(self, m, n):
        super().__init__()
        self.m, self.n = m, n

    ctrl_size = property(lambda self: self.m)
    state_size = property(lambda self: self.n)

    def f_func(self, X):
        return X * 0

    def g_func(self, X):
        return X.unsqueeze(-1) * X.new_zeros(*X.shape, self.m)


def isleaf(x):
    return x.grad_fn is None


@contextmanager
def variable_required_grad(x):
    """`with variable_required_grad(x) as xg:` -- x (or, for a non-leaf, a detached copy) with requires_grad switched on
    for the duration; a leaf gets its previous flag back."""
    leaf = isleaf(x)
    was = x.requires_grad
    target = x if leaf else x.detach().clone()
    try:
        yield target.requires_grad_(True)
    finally:
        if leaf:
            x.requires_grad_(was)


def t_hessian(f, x, xp, grad_check=True):
    """Cross second derivative  H[i, j] = d^2 f(x, xp) / d x_i d xp_j  of a scalar two-argument function."""
    with variable_required_grad(x), variable_required_grad(xp):
        first = torch.autograd.grad(f(x, xp), x, create_graph=True)[0]
        return t_jac(first, xp)


def gradgradcheck(f2, x):
    """Numerical check of the second derivatives of f2(x, x') (first derivatives taken as correct)."""
    xp = x.detach().clone()
    with variable_required_grad(x), variable_required_grad(xp):
        for i in range(x.shape[0]):
            torch.autograd.gradcheck(lambda xt, i=i: torch.autograd.grad(f2(x, xt), x, create_graph=True)[0][i], xp)


def epsilon(i, interpolate={0: 1, 1000: 0.01}):
    """Log-linear interpolation between two (step, value) anchors (exploration schedule)."""
    (s0, v0), (s1, v1) = list(interpolate.items())
    frac = (i - s0) / (s1 - s0)
    return math.exp(math.log(v0) + frac * (math.log(v1) - math.log(v0)))


def get_affine_terms(func, x):
    """(a, b) with func(x') = a . x' + b for an affine scalar `func`: the gradient at x and the remainder."""
    with variable_required_grad(x):
        val = func(x)
        a = torch.autograd.grad(val, x, create_graph=True)[0]
    with torch.no_grad():
        b = val - a @ x
    return a, b


def get_quadratic_terms(func, x):
    """(Q, p, r) with func(x') = x'.Q x' + p . x' + r for a quadratic scalar `func`: Q = Jacobian of the gradient / 2
    (NOT symmetrised, as the reference), then the remainders."""
    with variable_required_grad(x):
        val = func(x)
        grad = torch.autograd.grad(val, x, create_graph=True)[0]
        Q = t_jac(grad, x) / 2
    with torch.no_grad():
        p = grad - 2 * Q @ x
        r = val - x @ Q @ x - p @ x
    return Q, p, r


def clip(x, min_, max_):
    return torch.max(torch.min(x, max_), min_)


def random_psd(m):
    M = torch.rand(m, m)
    return M @ M.T


def normalize_radians(theta):
    return (theta + math.pi) % (2 * math.pi) - math.pi


def gitdescribe(f):
    out = subprocess.run(["git", "describe", "--always"], cwd=os.path.dirname(f) or ".", stdout=subprocess.PIPE)
    return out.stdout.decode("utf-8").strip()


def ensuredirs(fpath):
    d = os.path.dirname(fpath)
    if d and not os.path.exists(d):
        os.makedirs(d)
    return fpath


def quadratic_terms_by_autograd(cbc2, x, u):
    """cbc2.py:7-23 literally: `cbc2` is any callable u -> object with `.mean(x)` (affine in u) and `.knl(x, x)`
    (quadratic in u), both differentiable in u by autograd.  Returns ((mean_A, mean_b), (k_Q, k_p, k_r), mean(u),
    var(u)).  This is the fallback `cbc2.cbc2_quadratic_terms` uses for callables that are not this package's
    expression trees."""
    def mean(up):
        return cbc2(up).mean(x)

    def var(up):
        return cbc2(up).knl(x, x)

    mean_A, mean_b = get_affine_terms(mean, u)
    assert not torch.isnan(mean_A).any() and not torch.isnan(mean_b).any()
    k_Q, k_p, k_r = get_quadratic_terms(var, u)
    assert not (torch.isnan(k_Q).any() or torch.isnan(k_p).any() or torch.isnan(k_r).any())
    return (mean_A, mean_b), (k_Q, k_p, k_r), mean(u), var(u)
