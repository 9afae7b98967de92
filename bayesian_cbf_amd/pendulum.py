"""Mirror of the hot-path pieces of bayes_cbf/pendulum.py: the ground-truth pendulum model (:82-130) and the
rel-degree-2 radial barrier (:643-696) whose condition `cbc(u)` goes through the jet kernel + closed-form terms
(`cbc2.RelDeg2Safety`).  Controllers / plotting of the reference module are out of scope."""
import math

import torch

from .cbc2 import RelDeg2Safety


class PendulumDynamicsModel:
    """theta'' = -(g/l) sin(theta) + u / (m l):  f(x) = [omega, -(g/l) sin theta],  g(x) = [0, 1/(m l)]'  (:106-130)."""
    ground_truth = True

    def __init__(self, m=1, n=2, mass=1, gravity=10, length=1, deterministic=True, model_noise=0,
                 dtype=torch.get_default_dtype()):
        self.m, self.n, self.mass, self.gravity, self.length, self.dtype = m, n, mass, gravity, length, dtype

    def to(self, dtype):
        self.dtype = dtype

    @property
    def ctrl_size(self):
        return self.m

    @property
    def state_size(self):
        return self.n

    def f_func(self, X):
        theta, omega = X[..., 0:1], X[..., 1:2]
        return torch.cat([omega, -(self.gravity / self.length) * torch.sin(theta)], dim=-1)

    def g_func(self, x):
        gx = torch.tensor([[0.0], [1.0 / (self.mass * self.length)]], dtype=x.dtype, device=x.device)
        return gx.expand(*x.shape[:-1], 2, 1).clone() if x.ndim >= 2 else gx

    def F_func(self, X):
        return torch.cat([self.f_func(X).unsqueeze(-1), self.g_func(X)], dim=-1)


class RadialCBFRelDegree2(RelDeg2Safety):
    """h(x) = cos(delta_col) - cos(theta - theta_c)  (:675-696): keep the pendulum out of a cone around theta_c."""

    def __init__(self, model, cbf_col_gamma=1, _k_alpha=(1.0, 3.0), cbf_col_delta=math.pi / 8,
                 cbf_col_theta=math.pi / 4, theta_c=math.pi / 4, gamma_col=1, max_unsafe_prob=0.01,
                 delta_col=math.pi / 8, name="cbf-r2", dtype=torch.get_default_dtype()):
        self._model, self._max_unsafe_prob, self._k_alpha = model, max_unsafe_prob, list(_k_alpha)
        self.cbf_col_delta, self.cbf_col_theta, self.name, self.dtype = cbf_col_delta, cbf_col_theta, name, dtype

    k_alpha = property(lambda self: self._k_alpha)
    model = property(lambda self: self._model)
    max_unsafe_prob = property(lambda self: self._max_unsafe_prob)

    def cbf(self, x):
        return math.cos(self.cbf_col_delta) - torch.cos(x[..., 0] - self.cbf_col_theta)

    value = cbf

    def grad_cbf(self, X_in):
        X = X_in.unsqueeze(0) if X_in.ndim == 1 else X_in
        g = torch.cat((torch.sin(X[:, 0:1] - self.cbf_col_theta), X.new_zeros(X.shape[0], 1)), dim=-1)
        return g.squeeze(0) if X_in.ndim == 1 else g

    def hess_cbf(self, x):
        """d grad_cbf / dx (what GradientGP differentiates through, gp_algebra.py:340-345)."""
        H = x.new_zeros(2, 2)
        H[0, 0] = torch.cos(x[0] - self.cbf_col_theta)
        return H
