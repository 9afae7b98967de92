import contextlib


class _Ctx(contextlib.ContextDecorator):
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


max_cg_iterations = _Ctx
lazily_evaluate_kernels = _Ctx
debug = _Ctx
