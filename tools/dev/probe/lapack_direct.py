"""Call LAPACK routines of scipy's bundled library directly (function pointers out of scipy.linalg.cython_lapack's capsules)."""
import ctypes, numpy as np
from scipy.linalg import cython_lapack as cl
ctypes.pythonapi.PyCapsule_GetPointer.restype = ctypes.c_void_p
ctypes.pythonapi.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
ctypes.pythonapi.PyCapsule_GetName.restype = ctypes.c_char_p
ctypes.pythonapi.PyCapsule_GetName.argtypes = [ctypes.py_object]
def fptr(name):
    cap = cl.__pyx_capi__[name]
    return ctypes.pythonapi.PyCapsule_GetPointer(cap, ctypes.pythonapi.PyCapsule_GetName(cap))
I = ctypes.c_int; D = ctypes.c_double; P = ctypes.POINTER
def dlahqr(H, Z, ilo, ihi):
    """H, Z Fortran-order arrays (modified in place); 0-based inclusive ilo, ihi."""
    n = H.shape[0]
    f = ctypes.CFUNCTYPE(None, P(I), P(I), P(I), P(I), P(I), P(D), P(I), P(D), P(D), P(I), P(I), P(D), P(I), P(I))(fptr('dlahqr'))
    wr = np.zeros(n); wi = np.zeros(n); info = I(0)
    t = I(1); nn = I(n); lo = I(ilo + 1); hi = I(ihi + 1)
    f(ctypes.byref(t), ctypes.byref(t), ctypes.byref(nn), ctypes.byref(lo), ctypes.byref(hi), H.ctypes.data_as(P(D)), ctypes.byref(nn),
      wr.ctypes.data_as(P(D)), wi.ctypes.data_as(P(D)), ctypes.byref(lo), ctypes.byref(hi), Z.ctypes.data_as(P(D)), ctypes.byref(nn), ctypes.byref(info))
    return wr, wi, info.value
def dlanv2(a, b, c, d):
    f = ctypes.CFUNCTYPE(None, *([P(D)] * 10))(fptr('dlanv2'))
    v = [D(x) for x in (a, b, c, d, 0, 0, 0, 0, 0, 0)]
    f(*[ctypes.byref(x) for x in v])
    return [x.value for x in v]
def dlarfg(n, alpha, x):
    f = ctypes.CFUNCTYPE(None, P(I), P(D), P(D), P(I), P(D))(fptr('dlarfg'))
    a = D(alpha); x = np.array(x, dtype=float); tau = D(0); one = I(1); nn = I(n)
    f(ctypes.byref(nn), ctypes.byref(a), x.ctypes.data_as(P(D)), ctypes.byref(one), ctypes.byref(tau))
    return a.value, tau.value, x
