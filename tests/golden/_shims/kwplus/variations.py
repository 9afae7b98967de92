import itertools


class kwvariations(list):
    pass


def expand_variations(d):
    keys = list(d)
    vals = [v if isinstance(v, kwvariations) else [v] for v in d.values()]
    return [dict(zip(keys, combo)) for combo in itertools.product(*vals)]
