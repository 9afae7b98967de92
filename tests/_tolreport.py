"""Comparison helpers shared by the GPU parity tests.  With BCBF_TOL_REPORT=<file> set, every comparison also appends one
JSON line {test, what, err_over_scale, rtol} to that file, so the measured worst deviation behind every asserted
tolerance can be listed (`tools/tol_report.py`) -- the evidence for holding fp32 outputs to north_star's 1e-3."""
import json
import os

import numpy as np

_REPORT = os.environ.get("BCBF_TOL_REPORT")


def _record(what, ratio, rtol):
    if not _REPORT:
        return
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    with open(_REPORT, "a") as f:
        f.write(json.dumps({"test": test, "what": what, "err_over_scale": float(ratio), "rtol": float(rtol)}) + "\n")


def rel_close(actual, desired, rtol, scale=None, what=""):
    """max |actual - desired| <= rtol * scale (scale defaults to max |desired|)."""
    actual, desired = np.asarray(actual), np.asarray(desired)
    sc = np.abs(desired).max() if scale is None else scale
    err = np.abs(actual - desired).max() if actual.size else 0.0
    _record(what, err / max(sc, 1e-300), rtol)
    assert err <= rtol * max(sc, 1e-300), "%s: max abs err %.3e > %.1e * scale %.3e" % (what, err, rtol, sc)


def all_close(actual, desired, rtol, atol, what=""):
    """numpy.testing.assert_allclose with the worst |a - d| / (atol + rtol |d|) * rtol recorded."""
    actual, desired = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
    if actual.size:
        ratio = float((np.abs(actual - desired) / (atol + rtol * np.abs(desired))).max())
        _record(what, ratio * rtol, rtol)
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=atol, err_msg=what)
