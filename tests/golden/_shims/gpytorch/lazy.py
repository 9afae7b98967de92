class LazyTensor:  # names only: the lazy-tensor algebra is used by fit()/predict(), never by the oracle
    pass


class KroneckerProductLazyTensor(LazyTensor):
    pass


class BlockDiagLazyTensor(LazyTensor):
    pass


class InterpolatedLazyTensor(LazyTensor):
    pass


class NonLazyTensor(LazyTensor):
    pass


def lazify(x):
    raise NotImplementedError


def cat(*a, **k):
    raise NotImplementedError
