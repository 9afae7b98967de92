from . import memoize  # noqa
