import torch
from torch import nn
import torch.nn.functional as F


class _Evaluated:
    """Stands in for a LazyTensor: only .evaluate() is used on the prediction path."""
    def __init__(self, t):
        self._t = t

    def evaluate(self):
        return self._t


class Kernel(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()

    def __call__(self, x1, x2=None, **params):
        if x2 is None:
            x2 = x1
        out = self.forward(x1, x2, **params)
        return out if isinstance(out, _Evaluated) else _Evaluated(out)

    def __add__(self, other):
        return AdditiveKernel(self, other)


class AdditiveKernel(Kernel):
    def __init__(self, k1, k2):
        super().__init__()
        self.kernels = nn.ModuleList([k1, k2])

    def forward(self, x1, x2, **params):
        return self.kernels[0](x1, x2).evaluate() + self.kernels[1](x1, x2).evaluate()


class RBFKernel(Kernel):
    def __init__(self, ard_num_dims=None, lengthscale_prior=None, **kwargs):
        super().__init__()
        d = 1 if ard_num_dims is None else ard_num_dims
        self.raw_lengthscale = nn.Parameter(torch.zeros(1, d))

    @property
    def lengthscale(self):
        return F.softplus(self.raw_lengthscale)

    def forward(self, x1, x2, **params):
        a = x1 / self.lengthscale
        b = x2 / self.lengthscale
        d2 = ((a.unsqueeze(-2) - b.unsqueeze(-3)) ** 2).sum(-1)
        return torch.exp(-0.5 * d2)


class LinearKernel(Kernel):
    def __init__(self, **kwargs):
        super().__init__()
        self.raw_variance = nn.Parameter(torch.zeros(1, 1))

    @property
    def variance(self):
        return F.softplus(self.raw_variance)

    def forward(self, x1, x2, **params):
        return self.variance * (x1 @ x2.transpose(-2, -1))


class ScaleKernel(Kernel):
    def __init__(self, base_kernel, **kwargs):
        super().__init__()
        self.base_kernel = base_kernel
        self.raw_outputscale = nn.Parameter(torch.zeros(()))

    @property
    def outputscale(self):
        return F.softplus(self.raw_outputscale)

    def forward(self, x1, x2, **params):
        return self.outputscale * self.base_kernel(x1, x2).evaluate()


class IndexKernel(Kernel):
    def __init__(self, num_tasks, rank=1, prior=None, **kwargs):
        super().__init__()
        self.covar_factor = nn.Parameter(torch.randn(num_tasks, rank))
        self.raw_var = nn.Parameter(torch.randn(num_tasks))

    @property
    def var(self):
        return F.softplus(self.raw_var)

    @property
    def covar_matrix(self):
        return _Evaluated(self.covar_factor @ self.covar_factor.t() + torch.diag(self.var))

    def forward(self, i1, i2, **params):
        return self.covar_matrix.evaluate()[i1][:, i2]


class MultitaskKernel(Kernel):
    pass
