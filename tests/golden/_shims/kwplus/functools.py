from functools import partial


def recpartial(func, kw):
    """partial() with dotted keys reaching into nested partial keywords."""
    direct = {k: v for k, v in kw.items() if '.' not in k}
    nested = {}
    for k, v in kw.items():
        if '.' in k:
            head, rest = k.split('.', 1)
            nested.setdefault(head, {})[rest] = v
    for head, sub in nested.items():
        inner = func.keywords[head] if isinstance(func, partial) and head in func.keywords else None
        if inner is None:
            raise KeyError(head)
        direct[head] = recpartial(inner, sub)
    return partial(func, **direct)
