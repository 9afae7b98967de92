"""Minimal mirror of the reference's GP views (bayes_cbf/gp_algebra.py:70-106, 258-315).

The reference builds an expression tree over these objects and differentiates it with autograd to
obtain constraint terms; here the terms come from closed-form kernels (`ops.cbc_terms`,
`ops.cbc_socp`), so only the leaf types that user code touches are kept: `GaussianProcess`
(mean / knl / covar with registered cross-covariances) and `DeterministicGP`."""


class GaussianProcessBase:
    pass


class DeterministicGP(GaussianProcessBase):
    def __init__(self, mean, shape, name="{mean}"):
        self._mean, self._shape = mean, shape
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
    def __init__(self, mean, knl, shape, assume_independence=False, name="{mean}"):
        self._mean, self._knl, self._shape = mean, knl, shape
        self._covars = dict()
        self.register_covar(self, self.knl)
        self.assume_independence = assume_independence
        self._name = name.format(mean=mean)

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
        return x.new_zeros(max(self.shape), max(Z.shape))

    def register_covar(self, gp, covar_func):
        """One function for both directions, as the reference does (gp_algebra.py:306-309)."""
        self._covars[id(gp)] = covar_func
        gp._covars[id(self)] = covar_func
