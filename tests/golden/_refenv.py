"""Bootstrap for importing the reference (/root/reference, read-only) in the build container.

Only used by the golden-vector generators in this directory.  Puts the third-party
test doubles of `_shims/` and the reference on sys.path, provides
`torch.utils.tensorboard.SummaryWriter` (import-time name, reference misc.py:17) and
maps the removed `torch.eig` onto `torch.linalg.eig` (reference gp_algebra.py:385,389).
"""
import os
import sys
import types

REFERENCE = os.environ.get("BCBF_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REFERENCE, "bayes_cbf"))


def setup():
    if not available():
        raise RuntimeError("reference tree not found at %s (golden vectors are generated in the "
                           "build container only)" % REFERENCE)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    for p in (os.path.join(HERE, "_shims"), REFERENCE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    if "torch.utils.tensorboard" not in sys.modules:
        tb = types.ModuleType("torch.utils.tensorboard")

        class SummaryWriter:
            def __init__(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass

            def close(self):
                pass

        tb.SummaryWriter = SummaryWriter
        sys.modules["torch.utils.tensorboard"] = tb
        torch.utils.tensorboard = tb
    def eig(A, eigenvectors=False):          # torch.eig was removed in torch 1.13 (the stub left behind raises)
        w, V = torch.linalg.eig(A)
        return torch.stack([w.real, w.imag], dim=-1), V.real
    torch.eig = eig

    # the torch the reference was written for reported a failed factorisation as "... singular U." (the text
    # controllers.py:466,527 test for) and still had torch.symeig; restore both behaviours for the harness
    _chol = torch.linalg.cholesky

    def cholesky(A, upper=False):
        try:
            L = _chol(A)
        except RuntimeError as err:
            raise RuntimeError("cholesky_cpu: U(k,k) is zero, singular U. [%s]" % str(err).split(":")[0])
        return L.transpose(-1, -2) if upper else L

    torch.cholesky = cholesky

    def symeig(A, eigenvectors=False, upper=True):
        w, V = torch.linalg.eigh(A)
        return w, V

    torch.symeig = symeig
    return torch
