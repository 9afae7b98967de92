"""Stand-in for the author's kwplus helper (requirements.txt:16): config plumbing only."""
import inspect
from . import variations, functools  # noqa


def default_kw(func):
    try:
        sig = inspect.signature(func)
    except (TypeError, ValueError):
        return {}
    return {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
